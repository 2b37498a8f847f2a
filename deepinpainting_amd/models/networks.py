"""Network definitions around the IPSR layer — mirror of the reference's models/networks.py.

What must stay identical to the reference is the MODULE TREE (attribute names and nn.Sequential
positions), because checkpoints are plain state_dicts keyed by it (models/base_model.py:43-64), and the
factory signatures (`define_G`, `define_D`, `GANLoss`, `get_scheduler`, `init_weights`).  The four nets:

  netG  UnetGeneratorIPSR   refinement U-Net, 6->3 ch, blocks of [dilated 4x4 s2 conv, IN, 3x3 conv, IN];
                            the IPSR block (patch attention + InnerCos taps) sits at 32x32 (:187-209,281-366)
  netP  UnetGenerator       rough U-Net (pix2pix unet_256), 3->3 ch, Tanh output (:371-452)
  netD  NLayerDiscriminator 70x70 PatchGAN (:459-503)
  netF  PFDiscriminator     feature-patch discriminator on VGG "relu3_3" [B,256,32,32] (:504-520)

The convolutions run on PyTorch-ROCm (MIOpen); the patch-attention layer and the InnerCos taps inside
the IPSR block are the hand-written HIP path (IPSR_model / InnerCos / InnerCos2).
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init
from torch.optim import lr_scheduler

from .IPSR_model import IPSR_model
from .InnerCos import InnerCos
from .InnerCos2 import InnerCos2
from .fused import FusedSequential, cat_skip


# ----------------------------------------------------------------------------------------------------
# helpers (reference :20-78)
# ----------------------------------------------------------------------------------------------------
def get_norm_layer(norm_type='instance'):
    table = {
        'batch': functools.partial(nn.BatchNorm2d, affine=True),
        'instance': functools.partial(nn.InstanceNorm2d, affine=True),
        'none': None,
    }
    if norm_type not in table:
        raise NotImplementedError('normalization layer [%s] is not found' % norm_type)
    return table[norm_type]


def get_scheduler(optimizer, opt):
    policy = opt.lr_policy
    if policy == 'lambda':
        # linear decay to zero over niter_decay epochs after niter epochs at the base rate
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + 1 + opt.epoch_count - opt.niter) / float(opt.niter_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    if policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.niter, eta_min=0)
    # the reference RETURNS (not raises) the exception object here (:45); kept
    return NotImplementedError('learning rate policy [%s] is not implemented', policy)


def init_weights(net, init_type='normal', gain=0.02):
    def init_func(m):
        cls = m.__class__.__name__
        if hasattr(m, 'weight') and ('Conv' in cls or 'Linear' in cls):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif 'BatchNorm2d' in cls:
            init.normal_(m.weight.data, 1.0, gain)
            init.constant_(m.bias.data, 0.0)

    print('initialize network with %s' % init_type)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    if len(gpu_ids) > 0:
        assert (torch.cuda.is_available())
        net.cuda(gpu_ids[0])
    init_weights(net, init_type, gain=init_gain)
    return net


def define_G(input_nc, output_nc, ngf, which_model_netG, opt, mask_global, norm='batch', use_dropout=False,
             init_type='normal', gpu_ids=[], init_gain=0.02):
    """reference :81-103.  Returns (net, cosis_list, cosis_list2, ipsr_model): the three lists are the
    side channels through which the trainer reaches the IPSR layer objects (models/IPSR.py:51,155-164)."""
    norm_layer = get_norm_layer(norm_type=norm)
    cosis_list, cosis_list2, ipsr_model = [], [], []
    if which_model_netG == 'unet_256':
        netG = UnetGenerator(input_nc, output_nc, 8, ngf, norm_layer=norm_layer, use_dropout=use_dropout)
    elif which_model_netG == 'unet_ipsr':
        netG = UnetGeneratorIPSR(input_nc, output_nc, 8, opt, mask_global, ipsr_model, cosis_list, cosis_list2, ngf,
                                 norm_layer=norm_layer, use_dropout=use_dropout)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % which_model_netG)
    return init_net(netG, init_type, init_gain, gpu_ids), cosis_list, cosis_list2, ipsr_model


def define_D(input_nc, ndf, which_model_netD, n_layers_D=3, norm='batch', use_sigmoid=False, init_type='normal',
             gpu_ids=[], init_gain=0.02):
    norm_layer = get_norm_layer(norm_type=norm)
    if which_model_netD == 'basic':
        netD = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer, use_sigmoid=use_sigmoid)  # 3 is hard-coded (:112)
    elif which_model_netD == 'feature':
        netD = PFDiscriminator()
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % which_model_netD)
    return init_net(netD, init_type, init_gain, gpu_ids)


def print_network(net):
    num_params = sum(p.numel() for p in net.parameters())
    print(net)
    print('Total number of parameters: %d' % num_params)


# ----------------------------------------------------------------------------------------------------
# losses (reference :135-183)
# ----------------------------------------------------------------------------------------------------
class GANLoss(nn.Module):
    """Relativistic-average LSGAN.  NB the reference fills the "fake" target with real_label too (:167),
    so the target is 1 everywhere; reproduced."""

    def __init__(self, gan_type='wgan_gp', target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super(GANLoss, self).__init__()
        self.real_label = target_real_label
        self.fake_label = target_fake_label
        self.real_label_var = None
        self.fake_label_var = None
        self.Tensor = tensor
        if gan_type in ('wgan_gp', 'lsgan'):
            self.loss = nn.MSELoss()
        elif gan_type == 'vanilla':
            self.loss = nn.BCELoss()
        else:
            raise ValueError("GAN type [%s] not recognized." % gan_type)

    def get_target_tensor(self, input, target_is_real):
        attr = 'real_label_var' if target_is_real else 'fake_label_var'
        cur = getattr(self, attr)
        if cur is None or cur.numel() != input.numel() or cur.device != input.device:
            cur = torch.full(input.size(), self.real_label, dtype=input.dtype, device=input.device)
            setattr(self, attr, cur)
        return cur

    def __call__(self, y_pred_fake, y_pred, target_is_real):
        t = self.get_target_tensor(y_pred_fake, target_is_real)
        sign = 1.0 if target_is_real else -1.0
        a = torch.mean((y_pred - torch.mean(y_pred_fake) - sign * t) ** 2)
        b = torch.mean((y_pred_fake - torch.mean(y_pred) + sign * t) ** 2)
        return (a + b) / 2


# ----------------------------------------------------------------------------------------------------
# U-Net building blocks
# ----------------------------------------------------------------------------------------------------
def _cat_skip(block, x, head_act_done=False, tail_relu=False, cat_buf=None):
    """Shared forward of every skip block (:270-278, :358-366, :443-452).  `head_act_done`: the producer of x has already
    applied this level's leading in-place activation; `tail_relu`: the consumer's in-place ReLU is to be applied to the
    concatenated result here (both fused into kernels, see models/fused.py); `cat_buf`: the concatenated tensor, allocated by the
    producer of x with its skip half relu(x) already in place (fused._InstNormActSkip) — the level's last norm completes it."""
    if block.outermost:
        y = block.model(x, head_act_done=head_act_done)
        return torch.relu_(y) if tail_relu else y
    if tail_relu and isinstance(block.model, FusedSequential):
        y, done = block.model(x, head_act_done=head_act_done, cat_with=x, cat_buf=cat_buf)     # a level that ends in a norm concatenates by itself
        if done:
            return y
    else:
        y = block.model(x, head_act_done=head_act_done)
    h, w = x.size(2), x.size(3)
    if h != y.size(2) or w != y.size(3):
        y = F.interpolate(y, (h, w), mode='bilinear')
    return cat_skip(y, x, tail_relu)


def _block3_layers(outer_nc, inner_nc, input_nc, norm_layer):
    """The layer set every `_3` block is assembled from.  Conv geometry (reference :220-259):
       down : 4x4 stride-2 pad-3 DILATION-2 conv (input_nc -> input_nc), then 3x3 conv (input_nc -> inner_nc)
       up   : 3x3 transposed conv (2*inner_nc -> outer_nc), then 4x4 stride-2 transposed conv (outer_nc -> outer_nc)
    Construction order matches the reference so that seeded default inits consume the RNG identically."""
    L = {}
    L['downconv_3'] = nn.Conv2d(input_nc, inner_nc, kernel_size=3, stride=1, padding=1)
    L['downrelu_3'] = nn.LeakyReLU(0.2, True)
    L['downnorm_3'] = norm_layer(inner_nc, affine=True)
    L['uprelu_3'] = nn.ReLU(True)
    L['upnorm_3'] = norm_layer(outer_nc, affine=True)
    L['downconv'] = nn.Conv2d(input_nc, input_nc, kernel_size=4, stride=2, padding=3, dilation=2)
    L['downrelu'] = nn.LeakyReLU(0.2, True)
    L['downnorm'] = norm_layer(input_nc, affine=True)
    L['uprelu'] = nn.ReLU(True)
    L['upnorm'] = norm_layer(outer_nc, affine=True)
    return L


def _assemble3(L, outer_nc, inner_nc, submodule, outermost, innermost, use_dropout, mid_down=(), head_up=()):
    """Order the layers of a `_3` block.  `mid_down` goes between downconv_3 and downnorm_3, `head_up` in
    front of the up path (that is where the IPSR block inserts its attention layer and loss taps)."""
    if outermost:
        upconv_3 = nn.ConvTranspose2d(inner_nc * 2, outer_nc, kernel_size=3, stride=1, padding=1)
        return [L['downconv_3'], submodule, L['uprelu'], upconv_3]          # no Tanh on netG (:236-243)
    if innermost:
        upconv = nn.ConvTranspose2d(inner_nc, outer_nc, kernel_size=4, stride=2, padding=1)
        return [L['downrelu'], L['downconv'], L['uprelu'], upconv, L['upnorm']]
    upconv = nn.ConvTranspose2d(outer_nc, outer_nc, kernel_size=4, stride=2, padding=1)
    upconv_3 = nn.ConvTranspose2d(inner_nc * 2, outer_nc, kernel_size=3, stride=1, padding=1)
    down = [L['downrelu'], L['downconv'], L['downnorm'], L['downrelu_3'], L['downconv_3']] + list(mid_down) + [L['downnorm_3']]
    up = list(head_up) + [L['uprelu_3'], upconv_3, L['upnorm_3'], L['uprelu'], upconv, L['upnorm']]
    model = down + [submodule] + up
    if use_dropout:
        model = model + [nn.Dropout(0.5)]
    return model


class UnetSkipConnectionBlock_3(nn.Module):
    """reference :212-278."""

    def __init__(self, outer_nc, inner_nc, input_nc, submodule=None, outermost=False, innermost=False,
                 norm_layer=nn.BatchNorm2d, use_dropout=False):
        super(UnetSkipConnectionBlock_3, self).__init__()
        self.outermost = outermost
        if input_nc is None:
            input_nc = outer_nc
        L = _block3_layers(outer_nc, inner_nc, input_nc, norm_layer)
        self.model = FusedSequential(*_assemble3(L, outer_nc, inner_nc, submodule, outermost, innermost, use_dropout))

    def forward(self, x, head_act_done=False, tail_relu=False, cat_buf=None):
        return _cat_skip(self, x, head_act_done, tail_relu, cat_buf)


class IPSR(nn.Module):
    """The U-Net level that hosts the patch-attention layer (reference :281-366): after the 3x3 down conv
    (256->512 @ 32x32) come IPSR_model and the InnerCos tap, then the instance norm; InnerCos2 heads the
    up path.  The three objects are also appended to the caller's lists."""

    def __init__(self, outer_nc, inner_nc, opt, ipsr_model, cosis_list, cosis_list2, mask_global, input_nc,
                 submodule=None, outermost=False, innermost=False, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super(IPSR, self).__init__()
        self.outermost = outermost
        if input_nc is None:
            input_nc = outer_nc
        L = _block3_layers(outer_nc, inner_nc, input_nc, norm_layer)

        ipsr = IPSR_model(opt.threshold, opt.fixed_mask, opt.shift_sz, opt.stride, opt.mask_thred, opt.triple_weight)
        feat = ipsr.set_mask(mask_global, 3, opt.threshold)
        ipsr_model.append(ipsr)
        innerCos = InnerCos(strength=opt.strength, skip=opt.skip)
        innerCos.set_mask(mask_global, opt, feat_mask=feat)
        cosis_list.append(innerCos)
        innerCos2 = InnerCos2(strength=opt.strength, skip=opt.skip)
        innerCos2.set_mask(mask_global, opt, feat_mask=feat)
        cosis_list2.append(innerCos2)

        self.model = FusedSequential(*_assemble3(L, outer_nc, inner_nc, submodule, outermost, innermost, use_dropout,
                                               mid_down=(ipsr, innerCos), head_up=(innerCos2,)))

    def forward(self, x, head_act_done=False, tail_relu=False, cat_buf=None):
        return _cat_skip(self, x, head_act_done, tail_relu, cat_buf)


class UnetGeneratorIPSR(nn.Module):
    """netG (reference :187-209): innermost + (num_downs-5) + 1 plain 512-wide levels, the IPSR level,
    then 256/128/64-wide levels out to the image."""

    def __init__(self, input_nc, output_nc, num_downs, opt, mask_global, ipsr_model, cosis_list, cosis_list2, ngf=64,
                 norm_layer=nn.BatchNorm2d, use_dropout=False):
        super(UnetGeneratorIPSR, self).__init__()
        B3 = UnetSkipConnectionBlock_3
        blk = B3(ngf * 8, ngf * 8, input_nc=None, submodule=None, norm_layer=norm_layer, innermost=True)
        for _ in range(num_downs - 5):
            blk = B3(ngf * 8, ngf * 8, input_nc=None, submodule=blk, norm_layer=norm_layer, use_dropout=use_dropout)
        blk = B3(ngf * 8, ngf * 8, input_nc=None, submodule=blk, norm_layer=norm_layer, use_dropout=use_dropout)
        blk = IPSR(ngf * 4, ngf * 8, opt, ipsr_model, cosis_list, cosis_list2, mask_global, input_nc=None, submodule=blk,
                   norm_layer=norm_layer)
        blk = B3(ngf * 2, ngf * 4, input_nc=None, submodule=blk, norm_layer=norm_layer)
        blk = B3(ngf, ngf * 2, input_nc=None, submodule=blk, norm_layer=norm_layer)
        blk = B3(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True, norm_layer=norm_layer)
        self.model = blk

    def forward(self, input):
        return self.model(input)


class UnetSkipConnectionBlock(nn.Module):
    """pix2pix U-Net level of netP (reference :395-452): 4x4 stride-2 conv down, 4x4 stride-2 transposed conv up."""

    def __init__(self, outer_nc, inner_nc, input_nc, submodule=None, outermost=False, innermost=False,
                 norm_layer=nn.BatchNorm2d, use_dropout=False):
        super(UnetSkipConnectionBlock, self).__init__()
        self.outermost = outermost
        if input_nc is None:
            input_nc = outer_nc
        downconv = nn.Conv2d(input_nc, inner_nc, kernel_size=4, stride=2, padding=1)
        downrelu = nn.LeakyReLU(0.2, True)
        downnorm = norm_layer(inner_nc, affine=True)
        uprelu = nn.ReLU(True)
        upnorm = norm_layer(outer_nc, affine=True)
        up_in = inner_nc if innermost else inner_nc * 2
        upconv = nn.ConvTranspose2d(up_in, outer_nc, kernel_size=4, stride=2, padding=1)
        if outermost:
            model = [downconv, submodule, uprelu, upconv, nn.Tanh()]
        elif innermost:
            model = [downrelu, downconv, uprelu, upconv, upnorm]
        else:
            model = [downrelu, downconv, downnorm, submodule, uprelu, upconv, upnorm]
            if use_dropout:
                model.append(nn.Dropout(0.5))
        self.model = FusedSequential(*model)

    def forward(self, x, head_act_done=False, tail_relu=False, cat_buf=None):
        return _cat_skip(self, x, head_act_done, tail_relu, cat_buf)


class UnetGenerator(nn.Module):
    """netP (reference :371-388)."""

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super(UnetGenerator, self).__init__()
        Bk = UnetSkipConnectionBlock
        blk = Bk(ngf * 8, ngf * 8, input_nc=None, submodule=None, norm_layer=norm_layer, innermost=True)
        for _ in range(num_downs - 5):
            blk = Bk(ngf * 8, ngf * 8, input_nc=None, submodule=blk, norm_layer=norm_layer, use_dropout=use_dropout)
        for mult in (4, 2, 1):
            blk = Bk(ngf * mult, ngf * mult * 2, input_nc=None, submodule=blk, norm_layer=norm_layer)
        blk = Bk(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True, norm_layer=norm_layer)
        self.model = blk

    def forward(self, input):
        return self.model(input)


# ----------------------------------------------------------------------------------------------------
# discriminators
# ----------------------------------------------------------------------------------------------------
class NLayerDiscriminator(nn.Module):
    """PatchGAN (reference :459-503): k4 convs 3->64 s2 | 64->128 s2 IN | 128->256 s2 IN | 256->512 s1 IN | 512->1 s1."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False):
        super(NLayerDiscriminator, self).__init__()
        func = norm_layer.func if type(norm_layer) == functools.partial else norm_layer
        use_bias = func == nn.InstanceNorm2d
        kw, padw = 4, 1
        seq = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        mult = 1
        for n in range(1, n_layers + 1):
            prev, mult = mult, min(2 ** n, 8)
            stride = 2 if n < n_layers else 1
            seq += [nn.Conv2d(ndf * prev, ndf * mult, kernel_size=kw, stride=stride, padding=padw, bias=use_bias),
                    norm_layer(ndf * mult), nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv2d(ndf * mult, 1, kernel_size=kw, stride=1, padding=padw)]
        if use_sigmoid:
            seq += [nn.Sigmoid()]
        self.model = FusedSequential(*seq)

    def forward(self, input):
        return self.model(input)


class PFDiscriminator(nn.Module):
    """Feature-patch discriminator (reference :504-520) on the 256-channel VGG map: 3 stride-2 k4 convs to [B,512,4,4]."""

    def __init__(self):
        super(PFDiscriminator, self).__init__()
        self.model = FusedSequential(
            nn.Conv2d(256, 512, kernel_size=4, stride=2, padding=1),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(512, 512, kernel_size=4, stride=2, padding=1),
            nn.InstanceNorm2d(512),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(512, 512, kernel_size=4, stride=2, padding=1),
        )

    def forward(self, input):
        return self.model(input)
