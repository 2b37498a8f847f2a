"""BaseModel — mirror of the reference's models/base_model.py: option plumbing, per-net state_dict
checkpoints `{epoch}_net_{label}.pt` under `{checkpoints_dir}/{name}` (:43-64), LR stepping (:66-70)."""
import os

import torch


class BaseModel():
    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda', opt.gpu_ids[0]) if len(opt.gpu_ids) > 0 else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        os.makedirs(self.save_dir, exist_ok=True)

    def Tensor(self, *size):
        """fp32 buffer on the model's device (the reference aliases torch.cuda.FloatTensor, :12)."""
        return torch.empty(*size, dtype=torch.float32, device=self.device)

    # ---- hooks a concrete trainer overrides (reference :21-41: all of them are no-ops there too) ----
    def set_input(self, input):
        self.input = input

    def get_current_visuals(self):
        return self.input

    def get_current_errors(self):
        return {}

    def _noop(self, *args, **kwargs):
        return None

    forward = test = get_image_paths = optimize_parameters = save = _noop

    def _checkpoint_path(self, network_label, epoch_label):
        """`{epoch}_net_{label}.pt` under the run directory — the reference's file naming (:48,61)."""
        return os.path.join(self.save_dir, '%s_net_%s.pt' % (epoch_label, network_label))

    def save_network(self, network, network_label, epoch_label, gpu_ids):
        os.makedirs(self.save_dir, exist_ok=True)
        # the reference moves the net to the CPU and back (:56-58); saving a CPU copy of the state_dict
        # writes the same file without disturbing the resident parameters
        cpu_state = {key: value.detach().cpu() for key, value in network.state_dict().items()}
        torch.save(cpu_state, self._checkpoint_path(network_label, epoch_label))

    def load_network(self, network, network_label, epoch_label):
        state = torch.load(self._checkpoint_path(network_label, epoch_label), map_location=self.device, weights_only=True)
        network.load_state_dict(state)

    def update_learning_rate(self):
        """One scheduler tick per optimizer, then report the generator's rate like the reference (:66-70)."""
        for sched in self.schedulers:
            sched.step()
        print('learning rate = %.7f' % self.optimizers[0].param_groups[0]['lr'])
