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

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def test(self):
        pass

    def get_image_paths(self):
        pass

    def optimize_parameters(self):
        pass

    def get_current_visuals(self):
        return self.input

    def get_current_errors(self):
        return {}

    def save(self, label):
        pass

    def save_network(self, network, network_label, epoch_label, gpu_ids):
        os.makedirs(self.save_dir, exist_ok=True)
        save_path = os.path.join(self.save_dir, '%s_net_%s.pt' % (epoch_label, network_label))
        # the reference moves the net to the CPU and back (:56-58); saving a CPU copy of the state_dict
        # writes the same file without disturbing the resident parameters
        torch.save({k: v.detach().cpu() for k, v in network.state_dict().items()}, save_path)

    def load_network(self, network, network_label, epoch_label):
        save_path = os.path.join(self.save_dir, '%s_net_%s.pt' % (epoch_label, network_label))
        network.load_state_dict(torch.load(save_path, map_location=self.device))

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        print('learning rate = %.7f' % lr)
