"""IPSR — the two-stage inpainting GAN trainer, mirror of the reference's models/IPSR.py.

Same public surface: initialize / set_input / set_latent_mask / set_ref_latent /
set_gt_latent / forward / test / backward_D / backward_G / optimize_parameters / get_loss /
get_current_errors / get_current_visuals / get_error / save / load / update_learning_rate /
set_isTrain / set_isVal, and the attributes callers read (mask_global, netG/netP/netD/netF, optimizers,
schedulers).  The step order of optimize_parameters is the reference's (:267-278), including its quirks:

  * `forward()` zeroes the hole of input_A IN PLACE through `real_A.data.masked_fill_` (real_A aliases
    input_A, :169,174), so netG sees the zero-filled — not the mean-filled — image in channels 3..5;
  * the InnerCos losses enter loss_G as DETACHED leaves (:255-263): they change the logged value only;
  * GANLoss' target is 1 for "fake" too (networks.py:167).

What is new (none of it changes results):
  * one cal_feat_mask per set_input instead of three (:155-158 computes the same pyramid 3x);
  * VGG(input_B) of set_gt_latent is reused by backward_D (:187 vs :213 recompute the same features) —
    switch off with `opt.strict_reference = True`;
  * the VGG pass over the generated image stops after slice 3: only its relu3_3 is ever read (:192,218), the three
    512-channel convolutions of slice 4 were computed and dropped (same switch);
  * backward_G runs with the discriminators' parameters frozen: the gradients it would leave in netD / netF are cleared
    unused by the next optimize_parameters (:269-270) (same switch);
  * backward_D sends the fake and the real batch through netD (and netF) in ONE pass of 2B samples (`opt.batch_disc`, default on;
    the reference makes two passes, :196-199): same predictions, the weight gradients become one sum over 2B samples instead of
    two sums added by autograd — fp32 summation order only (+2 % images/s).  Only while the discriminators hold no batch-statistics
    module (`opt.norm='batch'` puts BatchNorm2d there: the two passes are kept, `_disc_is_per_sample`);
  * bias / InstanceNorm / activation between the convolutions run as fused HIP kernels (models/fused.py), the frozen
    VGG's bias / ReLU / max-pool likewise (models/vgg16.py) — same values up to fp32 rounding;
  * data parallelism: with torch.distributed initialised (one process per GPU, RCCL) the gradients are
    bucket-all-reduced by deepinpainting_amd.dist.GradBucketReducer, overlapped with the backward.
"""
import os
from collections import OrderedDict

import torch

from . import networks
from .base_model import BaseModel
from .vgg16 import Vgg16
from .. import dist as ipsr_dist


class IPSR(BaseModel):
    def name(self):
        return 'IPSRModel'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self.opt = opt
        self.isTrain = opt.isTrain
        self.strict_reference = bool(getattr(opt, 'strict_reference', False))
        self.batch_vgg = bool(getattr(opt, 'batch_vgg', False))
        self.batch_disc = bool(getattr(opt, 'batch_disc', True))       # backward_D: fake + real through netD / netF in one 2B pass
        # BASELINE config 5: convolutions under bf16 autocast (CDNA4 bf16 MFMA); the IPSR layer, the InnerCos taps and all
        # losses stay fp32.  Off by default — the reference is fp32.
        self.amp_bf16 = bool(getattr(opt, 'amp_bf16', False))
        # arithmetic of the Winograd convolution engines (models/hipconv.py, ops.MATH_CODE): fp32 operands on the fp32 matrix cores by
        # default (the reference's arithmetic); "bf16x6" (fp32-accurate, split operands on the bf16 matrix cores) / "bf16x3" are opt-in.
        # Under amp_bf16 the engines read / write bf16 activations and multiply split-bf16 operands (`conv_math_bf16`, default "bf16x3").
        from . import hipconv
        hipconv.set_conv_math(fp32=getattr(opt, 'conv_math', 'fp32'), bf16=getattr(opt, 'conv_math_bf16', 'bf16x3'))
        # conv-bias + InstanceNorm + activation in one HIP kernel each way (models/fused.py); False = plain torch modules
        networks.FusedSequential.enabled = bool(getattr(opt, 'fused_norm_act', True))

        # The reference always runs ImageNet-pretrained VGG16 features (models/vgg16.py:9 downloads them).  There is no
        # network here, so the weights come from a local file (opt.vgg16_weights / IPSR_VGG16_WEIGHTS); a seeded-random
        # VGG — patch matching, both InnerCos targets and netF would all run on random features — is only built when the
        # caller says so explicitly (opt.allow_random_vgg, or IPSR_ALLOW_RANDOM_VGG=1: bench.py and the tests do).
        weights = getattr(opt, 'vgg16_weights', None) or os.environ.get('IPSR_VGG16_WEIGHTS')
        allow_random = bool(getattr(opt, 'allow_random_vgg', False)) or os.environ.get('IPSR_ALLOW_RANDOM_VGG', '0') == '1'
        if not weights and not allow_random:
            raise RuntimeError(
                "IPSR: no VGG16 weights given.  The reference uses torchvision's ImageNet-pretrained vgg16 "
                "(models/vgg16.py:9); pass a local copy of that state_dict as opt.vgg16_weights or IPSR_VGG16_WEIGHTS=/path/to/vgg16.pth. "
                "For synthetic benchmarks / tests set opt.allow_random_vgg = True (or IPSR_ALLOW_RANDOM_VGG=1) to build a "
                "seeded-random VGG16 instead.")
        self.vgg = Vgg16(requires_grad=False, weights_path=weights).to(self.device)
        self.vgg.eval()
        if not self.vgg.pretrained and not getattr(opt, 'quiet', False):
            print('WARNING: VGG16 feature extractor is seeded-random (allow_random_vgg) — not the pretrained net the reference uses')

        fs = opt.fineSize
        self.input_A = self.Tensor(opt.batchSize, opt.input_nc, fs, fs)
        self.input_B = self.Tensor(opt.batchSize, opt.output_nc, fs, fs)
        self.input_ref = self.Tensor(opt.batchSize, opt.output_nc, fs, fs)

        # one mask for the whole (local) batch; default = centred square shrunk by `overlap` (:36-41)
        self.mask_global = torch.zeros(1, 1, fs, fs, dtype=torch.bool, device=self.device)
        lo, hi = int(fs / 4) + opt.overlap, int(fs / 2) + int(fs / 4) - opt.overlap
        self.mask_global[:, :, lo:hi, lo:hi] = 1
        self.mask_type = opt.mask_type
        self.gMask_opts = {}
        self.use_gpu = len(opt.gpu_ids) > 0

        self.netG, self.Cosis_list, self.Cosis_list2, self.CSA_model = networks.define_G(
            opt.input_nc_g, opt.output_nc, opt.ngf, opt.which_model_netG, opt, self.mask_global, opt.norm,
            opt.use_dropout, opt.init_type, self.gpu_ids, opt.init_gain)
        self.netP, _, _, _ = networks.define_G(
            opt.input_nc, opt.output_nc, opt.ngf, opt.which_model_netP, opt, self.mask_global, opt.norm,
            opt.use_dropout, opt.init_type, self.gpu_ids, opt.init_gain)
        # BASELINE config 5 names "bf16 MFMA for patch-corr + convs": with amp_bf16 the layer's correlation runs on the bf16
        # MFMA kernel too (opt.bf16_corr = False keeps it fp32); every other part of the layer stays fp32
        self.CSA_model[0].corr_bf16 = self.amp_bf16 and bool(getattr(opt, 'bf16_corr', True)) and self.device.type == 'cuda'
        if self.isTrain:
            use_sigmoid = opt.gan_type == 'vanilla'
            self.netD = networks.define_D(opt.input_nc, opt.ndf, opt.which_model_netD, opt.n_layers_D, opt.norm,
                                          use_sigmoid, opt.init_type, self.gpu_ids, opt.init_gain)
            self.netF = networks.define_D(opt.input_nc, opt.ndf, opt.which_model_netF, opt.n_layers_D, opt.norm,
                                          use_sigmoid, opt.init_type, self.gpu_ids, opt.init_gain)

        if not self.isTrain or opt.continue_train:
            print('Loading pre-trained network!')
            self.load_network(self.netG, 'G', opt.which_epoch)
            self.load_network(self.netP, 'P', opt.which_epoch)
            if self.isTrain:
                self.load_network(self.netD, 'D', opt.which_epoch)
                self.load_network(self.netF, 'F', opt.which_epoch)

        self.criterionGAN = networks.GANLoss(gan_type=opt.gan_type)
        self.criterionL1 = torch.nn.L1Loss()
        self._gt_latent = None
        self._reducer_D = self._reducer_G = None

        if self.isTrain:
            self.old_lr = opt.lr
            # same optimizer as the reference (torch.optim.Adam(lr, betas=(beta1, 0.999)), models/IPSR.py:89-96); on the GPU
            # its single-pass fused implementation is used (the default multi-tensor one makes ~10 passes over the
            # 145 M parameters, 2.5 ms/step); `opt.fused_adam = False` restores the default
            fused = bool(getattr(opt, 'fused_adam', True)) and self.device.type == 'cuda'

            def mk(net):
                kw = dict(lr=opt.lr, betas=(opt.beta1, 0.999))
                if fused:
                    try:
                        return torch.optim.Adam(net.parameters(), fused=True, **kw)
                    except (TypeError, RuntimeError):
                        pass
                return torch.optim.Adam(net.parameters(), **kw)
            self.optimizer_G, self.optimizer_P = mk(self.netG), mk(self.netP)
            self.optimizer_D, self.optimizer_F = mk(self.netD), mk(self.netF)
            self.optimizers = [self.optimizer_G, self.optimizer_P, self.optimizer_D, self.optimizer_F]
            self.schedulers = [networks.get_scheduler(o, opt) for o in self.optimizers]

            if torch.distributed.is_available() and torch.distributed.is_initialized() \
                    and torch.distributed.get_world_size() > 1:
                for net in (self.netG, self.netP, self.netD, self.netF):
                    ipsr_dist.broadcast_module(net, src=0)
                bucket = int(getattr(opt, 'ddp_bucket_mb', 64)) << 20
                self._reducer_D = ipsr_dist.GradBucketReducer([self.netD, self.netF], bucket_bytes=bucket)
                self._reducer_G = ipsr_dist.GradBucketReducer([self.netG, self.netP], bucket_bytes=bucket)

            if not getattr(opt, 'quiet', False):
                print('---------- Networks initialized -------------')
                for net in (self.netG, self.netP, self.netD, self.netF):
                    networks.print_network(net)
                print('-----------------------------------------------')

    def _amp(self):
        return torch.autocast(device_type='cuda', dtype=torch.bfloat16, enabled=self.amp_bf16 and self.device.type == 'cuda')

    def set_isTrain(self):
        self.isTrain = True

    def set_isVal(self):
        self.isTrain = False

    # ------------------------------------------------------------------------------------------------
    def set_input(self, input, mask, ref):
        """reference :120-152.  input [B,3,H,W] in [-1,1] (ground truth), mask [1,1,H,W] bool, ref [B,3,H,W].
        Extension: a [B,1,H,W] mask gives every sample its own hole (models/IPSR_model.py)."""
        self.input_A.resize_(input.size()).copy_(input)
        self.input_B.resize_(input.size()).copy_(input)
        self.input_ref.resize_(ref.size()).copy_(ref)
        self.image_paths = 0
        self._gt_latent = None

        if self.opt.mask_type == 'center':
            pass
        elif self.opt.mask_type == 'random':
            self.mask_global = mask.to(self.device)
        else:
            raise ValueError("Mask_type [%s] not recognized." % self.opt.mask_type)

        mg = self.mask_global
        self.ex_mask = mg.expand(mg.size(0), 3, mg.size(2), mg.size(3))
        self.inv_ex_mask = torch.add(torch.neg(self.ex_mask.float()), 1).bool()
        # fill the hole with the ImageNet channel means mapped to [-1,1] (:148-150)
        for ch, mean in enumerate((123.0, 104.0, 117.0)):
            self.input_A.narrow(1, ch, 1).masked_fill_(mg, 2 * mean / 255.0 - 1.0)
        self.set_latent_mask(mg, 3, self.opt.threshold)

    def set_latent_mask(self, mask_global, layer_to_last, threshold):
        """reference :155-158 — one feature-mask pyramid shared by the layer and both loss taps."""
        feat = self.CSA_model[0].set_mask(mask_global, layer_to_last, threshold)
        feat4 = feat[:, None] if feat.dim() == 3 else feat[None, None]
        shared = feat4 if (layer_to_last == 3 and threshold == self.opt.threshold) else None
        self.Cosis_list[0].set_mask(mask_global, self.opt, feat_mask=shared)
        self.Cosis_list2[0].set_mask(mask_global, self.opt, feat_mask=shared)

    def _vgg_ref_and_gt(self):
        """VGG features of the reference image and of the ground truth in ONE [2B] pass (both are known after
        set_input; the reference runs two [B] passes, models/IPSR.py:163,187).  Per-sample results are unchanged."""
        from .vgg16 import VggOutputs
        B = self.input_ref.size(0)
        with torch.no_grad(), self._amp():
            both = self.vgg(torch.cat((self.input_ref, self.input_B), 0))
        self._ref_latent_cache = VggOutputs(*[t[:B] for t in both])
        self._gt_latent = VggOutputs(*[t[B:] for t in both])

    def set_ref_latent(self):
        if self.batch_vgg and not self.strict_reference:
            self._vgg_ref_and_gt()
            self.ref_latent = self._ref_latent_cache
        else:
            with torch.no_grad(), self._amp():
                self.ref_latent = self.vgg(self.input_ref)
        self.CSA_model[0].set_ref(self.ref_latent)

    def set_gt_latent(self):
        if self._gt_latent is not None and self.batch_vgg and not self.strict_reference:
            gt_latent = self._gt_latent                  # computed together with the reference features
        else:
            with torch.no_grad(), self._amp():
                gt_latent = self.vgg(self.input_B)
        self._gt_latent = gt_latent
        self.Cosis_list[0].set_target(gt_latent.relu4_3)
        self.Cosis_list2[0].set_target(gt_latent.relu4_3)

    # ------------------------------------------------------------------------------------------------
    def _two_stage(self):
        self.real_A = self.input_A.to(self.device)                                   # alias of input_A
        with self._amp():
            self.fake_P = self.netP(self.real_A)
        self.fake_P = self.fake_P.float()
        self.un = self.fake_P.clone()
        self.Unknowregion = self.un.data.masked_fill_(self.inv_ex_mask, 0)           # rough result inside the hole
        self.knownregion = self.real_A.data.masked_fill_(self.ex_mask, 0)            # NB zeroes input_A's hole in place
        self.Syn = self.Unknowregion + self.knownregion
        self.Middle = torch.cat((self.Syn, self.input_A), 1)
        with self._amp():
            self.fake_B = self.netG(self.Middle)
        self.fake_B = self.fake_B.float()
        self.real_B = self.input_B.to(self.device)
        self.real_Ref = self.input_ref.to(self.device)

    def forward(self):
        self._two_stage()

    def test(self):
        self._two_stage()
        self.loss_IPSR = self.criterionGAN(self.real_B, self.fake_B, False)

    def get_loss(self):
        self.loss_valid = (self.criterionL1(self.fake_B, self.real_B) + self.criterionL1(self.fake_P, self.real_B)) * self.opt.lambda_A
        return OrderedDict([('GAN', self.loss_valid.data.item())])

    def _disc_is_per_sample(self):
        """The one-pass form of backward_D is the reference's arithmetic only while no module of netD / netF mixes samples:
        with `opt.norm='batch'` (still accepted by get_norm_layer, and define_D's default in the reference, networks.py:97-116) a
        BatchNorm2d would normalise fake and real with SHARED batch statistics and update its running statistics once instead of
        twice — different predictions, losses and gradients.  Then the reference's two passes are made."""
        ok = getattr(self, '_disc_per_sample', None)
        if ok is None:
            bn = torch.nn.modules.batchnorm._BatchNorm
            ok = not any(isinstance(m, bn) for net in (self.netD, self.netF) for m in net.modules())
            self._disc_per_sample = ok
        return ok

    def backward_D(self):
        fake_AB = self.fake_B
        with torch.no_grad(), self._amp():
            # only relu3_3 of the generated image is read (below and in backward_G): skip VGG slice 4 unless asked for
            # the reference's exact sequence
            self.gt_latent_fake = self.vgg(self.fake_B.data, last_slice=4 if self.strict_reference else 3)
            if self._gt_latent is not None and not self.strict_reference:
                self.gt_latent_real = self._gt_latent                               # same features as :213 recomputes
            else:
                self.gt_latent_real = self.vgg(self.input_B)
        real_AB = self.real_B

        with self._amp():
            if self.batch_disc and not self.strict_reference and self._disc_is_per_sample():
                # fake and real batch through each discriminator in ONE pass of 2B samples: InstanceNorm is per sample and the
                # convolutions per sample, so every prediction is the same number; the weight gradients become one sum over 2B
                # samples instead of two sums added by autograd (fp32 summation order, nothing else)
                nb = fake_AB.size(0)
                both = self.netD(torch.cat((fake_AB.detach(), real_AB), 0)).float()
                self.pred_fake, self.pred_real = both[:nb], both[nb:]
                both = self.netF(torch.cat((self.gt_latent_fake.relu3_3.detach(), self.gt_latent_real.relu3_3), 0)).float()
                self.pred_fake_F, self.pred_real_F = both[:nb], both[nb:]
            else:
                self.pred_fake = self.netD(fake_AB.detach()).float()
                self.pred_real = self.netD(real_AB).float()
                self.pred_fake_F = self.netF(self.gt_latent_fake.relu3_3.detach()).float()
                self.pred_real_F = self.netF(self.gt_latent_real.relu3_3).float()
        self.loss_D_fake = self.criterionGAN(self.pred_fake, self.pred_real, True)
        self.loss_F_fake = self.criterionGAN(self.pred_fake_F, self.pred_real_F, True)

        self.loss_D = self.loss_D_fake * 0.5 + self.loss_F_fake * 0.5
        if self._reducer_D is not None:
            self._reducer_D.arm()
        self.loss_D.backward()
        if self._reducer_D is not None:
            self._reducer_D.finish()

    def backward_G(self):
        # What loss_G.backward() deposits in netD / netF is never used: optimize_parameters() clears those gradients
        # before the next backward_D (:269-270).  With the discriminators' parameters frozen for this backward, autograd
        # skips their weight gradients and the whole graph of the two "real" branches and of the feature discriminator
        # (whose inputs do not depend on the generator); the generator's gradients are unchanged.
        frozen = [] if self.strict_reference else [p for net in (self.netD, self.netF) for p in net.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        try:
            self._backward_G()
        finally:
            for p in frozen:
                p.requires_grad_(True)

    def _backward_G(self):
        with self._amp():
            pred_fake = self.netD(self.fake_B).float()
            pred_fake_f = self.netF(self.gt_latent_fake.relu3_3).float()
            pred_real = self.netD(self.real_B).float()
            pred_real_F = self.netF(self.gt_latent_real.relu3_3).float()

        self.loss_G_GAN = self.criterionGAN(pred_fake, pred_real, False) + self.criterionGAN(pred_fake_f, pred_real_F, False)
        self.loss_G_L1 = (self.criterionL1(self.fake_B, self.real_B) + self.criterionL1(self.fake_P, self.real_B)) * self.opt.lambda_A
        self.loss_G = self.loss_G_L1 + self.loss_G_GAN * self.opt.gan_weight

        self.ng_loss_value = 0
        self.ng_loss_value2 = 0
        if self.opt.cosis:
            for gl in self.Cosis_list:
                self.ng_loss_value += gl.loss.detach()
            self.loss_G += self.ng_loss_value
            for gl in self.Cosis_list2:
                self.ng_loss_value2 += gl.loss.detach()
            self.loss_G += self.ng_loss_value2

        if self._reducer_G is not None:
            self._reducer_G.arm()
        self.loss_G.backward()
        if self._reducer_G is not None:
            self._reducer_G.finish()

    def optimize_parameters(self):
        self.forward()
        self.optimizer_D.zero_grad()
        self.optimizer_F.zero_grad()
        self.backward_D()
        self.optimizer_D.step()
        self.optimizer_F.step()
        self.optimizer_G.zero_grad()
        self.optimizer_P.zero_grad()
        self.backward_G()
        self.optimizer_G.step()
        self.optimizer_P.step()

    # ------------------------------------------------------------------------------------------------
    def get_current_errors(self):
        return OrderedDict([('G_GAN', self.loss_G_GAN.data.item()),
                            ('G_L1', self.loss_G_L1.data.item()),
                            ('D', self.loss_D_fake.data.item()),
                            ('F', self.loss_F_fake.data.item())])

    def get_current_visuals(self):
        return self.real_A.data, self.real_Ref.data, self.fake_B.data, self.fake_P.data, self.real_B.data

    def get_error(self):
        return self.loss_IPSR

    def save(self, epoch):
        for net, label in ((self.netG, 'G'), (self.netP, 'P'), (self.netD, 'D'), (self.netF, 'F')):
            self.save_network(net, label, epoch, self.gpu_ids)

    def load(self, epoch):
        self.load_network(self.netG, 'G', epoch)
        self.load_network(self.netP, 'P', epoch)
