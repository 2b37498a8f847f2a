"""Vgg16 feature extractor — mirror of the reference's models/vgg16.py WITHOUT torchvision and WITHOUT
any network access.

The reference builds torchvision's vgg16(pretrained=True).features (which downloads ImageNet weights)
and cuts it into four slices [0:5) [5:10) [10:17) [17:23) (:9-21).  Each of the first three slices ENDS
with its max-pool, so "relu3_3" is 256 ch @ H/8 and "relu4_3" is 512 ch @ H/8 — the map the IPSR layer
matches against.  Same slice names and in-slice indices here, so a torchvision `features.*` state_dict
can be loaded with `load_torchvision_state_dict`.  Weights: `IPSR_VGG16_WEIGHTS=/path/to/vgg16.pth`
(or the `weights_path` argument); without it the net is seeded-random (He normal) — "parity unpinned",
see DESIGN.md.
"""
import os
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision vgg16 'D' configuration up to features[22]
_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512]
_SLICES = ((0, 5), (5, 10), (10, 17), (17, 23))

VggOutputs = namedtuple("VggOutputs", ['relu1_2', 'relu2_2', 'relu3_3', 'relu4_3'])


def _feature_layers():
    layers, cin = [], 3
    for v in _CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return layers


class Vgg16(torch.nn.Module):
    def __init__(self, requires_grad=False, weights_path=None, seed=1234):
        super(Vgg16, self).__init__()
        feats = _feature_layers()
        assert len(feats) == 23
        for si, (lo, hi) in enumerate(_SLICES, start=1):
            seq = torch.nn.Sequential()
            for idx in range(lo, hi):
                seq.add_module(str(idx), feats[idx])
            setattr(self, 'slice%d' % si, seq)
        weights_path = weights_path or os.environ.get('IPSR_VGG16_WEIGHTS')
        if weights_path:
            self.load_torchvision_state_dict(torch.load(weights_path, map_location='cpu', weights_only=True))
            self.pretrained = True
        else:
            gen = torch.Generator().manual_seed(seed)
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    fan_out = m.weight.size(0) * m.weight.size(2) * m.weight.size(3)
                    with torch.no_grad():
                        m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * (2.0 / fan_out) ** 0.5)
                        m.bias.zero_()
            self.pretrained = False
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision_state_dict(self, sd):
        """Accepts torchvision's vgg16 state_dict ('features.N.weight') or this module's own."""
        own = self.state_dict()
        mapped = {}
        for k, v in sd.items():
            if k.startswith('features.'):
                idx = int(k.split('.')[1])
                for si, (lo, hi) in enumerate(_SLICES, start=1):
                    if lo <= idx < hi:
                        mapped['slice%d.%d.%s' % (si, idx, k.split('.')[2])] = v
            elif k in own:
                mapped[k] = v
        missing = [k for k in own if k not in mapped]
        if missing:
            raise RuntimeError("VGG16 weights file lacks %s" % missing[:4])
        self.load_state_dict(mapped)

    def _fused_slice(self, seq, x):
        """No-grad HIP path of one slice (fp32, or bf16 activations under autocast).  Per convolution:
          * Winograd F(4x4,3x3) (csrc/winograd.hip) where models/hipconv.py selects it, with the bias + ReLU (+ the 2x2
            max-pool when it follows) done in the kernel's output transform and the transformed filter cached per layer
            (the weights are frozen);
          * conv1_1 (3 input channels at full resolution): csrc/thin_conv.hip, bias + ReLU in the same pass;
          * otherwise the convolution without bias (MIOpen has none; PyTorch would add it in a separate pass), then ONE pass
            for bias + ReLU (+ pool) — ops.bias_act_ / ops.bias_relu_pool2.
        Both are value-identical to Conv2d(bias) -> ReLU(inplace) -> MaxPool2d up to the convolution's own rounding."""
        from .. import ops
        from . import hipconv
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                pool = i + 2 < len(mods) and isinstance(mods[i + 2], nn.MaxPool2d) and x.size(2) % 2 == 0 and x.size(3) % 2 == 0
                B, Cin, H, W = x.shape
                bf16 = hipconv._amp_bf16() or x.dtype == torch.bfloat16
                if (bf16 or not torch.is_autocast_enabled()) and not m.weight.requires_grad and \
                        hipconv.select(ops.CONV_FWD, B, Cin, H, W, m.out_channels, 3, 1, 1, 1, bf16) == "winograd":
                    math = hipconv._MATH["bf16" if bf16 else "fp32"]
                    key = (m.weight.data_ptr(), m.weight._version, x.device, math)
                    cache = getattr(m, "_ipsr_wino_filter", None)
                    valid = cache is not None and cache[0] == key
                    if not valid:
                        cache = (key, ops.winograd_filter_cache(ops.CONV_FWD, Cin, m.out_channels, x.device))
                        m._ipsr_wino_filter = cache
                    x = ops.conv3x3_winograd(ops.CONV_FWD, x.contiguous(), m.weight, (B, Cin, H, W), m.out_channels, bias=m.bias,
                                             epilogue="relu_pool" if pool else "relu", filter_cache=cache[1], filter_cache_valid=valid,
                                             math=math, out_dtype=torch.bfloat16 if bf16 else torch.float32)
                    i += 3 if pool else 2
                    continue
                if (bf16 or (x.dtype == torch.float32 and not torch.is_autocast_enabled())) and not pool and \
                        hipconv.select(ops.CONV_FWD, B, Cin, H, W, m.out_channels, 3, 1, 1, 1, bf16) == "thin":
                    x = ops.conv3x3_thin(ops.CONV_FWD, x.contiguous(), m.weight, (B, Cin, H, W), m.out_channels, bias=m.bias, relu=True,
                                         out_dtype=torch.bfloat16 if bf16 else torch.float32)                                        # conv1_1
                    i += 2
                    continue
                y = hipconv.conv_nobias(m, x)
                if pool:
                    x = ops.bias_relu_pool2(y, m.bias)
                    i += 3
                else:
                    x = ops.bias_act_(y, m.bias, "relu")
                    i += 2
            else:
                x = m(x)
                i += 1
        return x

    def forward(self, X, last_slice=4):
        """`last_slice` < 4 stops after that slice (later outputs are None): the trainer's pass over the generated image
        only feeds relu3_3 to the feature discriminator (models/IPSR.py:234,249 here; reference :192,218), so the three
        512-channel convolutions of slice 4 would be computed and thrown away."""
        fused = (X.is_cuda and X.dtype in (torch.float32, torch.bfloat16)
                 and not (torch.is_grad_enabled() and (X.requires_grad or self.slice1[0].weight.requires_grad)))
        if fused:
            X = X.contiguous()
            h1 = self._fused_slice(self.slice1, X)
            h2 = self._fused_slice(self.slice2, h1) if last_slice >= 2 else None
            h3 = self._fused_slice(self.slice3, h2) if last_slice >= 3 else None
            h4 = self._fused_slice(self.slice4, h3) if last_slice >= 4 else None
            return VggOutputs(h1, h2, h3, h4)
        h1 = self.slice1(X)
        h2 = self.slice2(h1) if last_slice >= 2 else None
        h3 = self.slice3(h2) if last_slice >= 3 else None
        h4 = self.slice4(h3) if last_slice >= 4 else None
        return VggOutputs(h1, h2, h3, h4)
