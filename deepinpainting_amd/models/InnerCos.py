"""InnerCos / InnerCos2 — feature-consistency loss taps (reference models/InnerCos.py, models/InnerCos2.py).

Pass-through modules: forward returns its input unchanged and stashes
    loss = MSE( (x * mask) * strength , target )
(InnerCos2 on the first 512 channels of the skip-concatenated tensor, InnerCos2.py:38).  The value is
computed by ONE fused HIP reduction (K9) instead of three element-wise passes + a reduction; it is an
autograd node (so the reference's `.backward()` method keeps working) although the trainer only ever
adds it detached (models/IPSR.py:258,262).
"""
import torch
import torch.nn as nn

from .. import ops
from ..util import util


class _InnerCosLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cuse, mask, target, strength):
        xf, tf = x.detach().float(), target.detach().float()       # fp32 whatever the surrounding autocast regime
        # The tap returns its INPUT (reference InnerCos.py:41), and the next module of the U-Net rewrites that tensor in place
        # (uprelu_3 after InnerCos2, models/networks.py:229): the backward must see the values the loss was computed on, so a
        # tensor that will take part in autograd is saved as a private copy (an alias would trip autograd's version check in
        # `InnerCos.backward()`; the reference's mul/MSE graph copies implicitly).  Under no_grad nothing is kept.
        keep = xf.clone() if (ctx.needs_input_grad[0] and xf.data_ptr() == x.data_ptr()) else xf
        ctx.save_for_backward(keep, mask, tf)
        ctx.cuse, ctx.strength, ctx.in_dtype = cuse, strength, x.dtype
        return ops.innercos_loss(xf, cuse, mask, tf, strength)

    @staticmethod
    def backward(ctx, grad_loss):
        x, mask, target = ctx.saved_tensors
        g = ops.innercos_loss_backward(x, ctx.cuse, mask, target, ctx.strength, grad_loss.float())
        return g.to(ctx.in_dtype), None, None, None, None


class InnerCos(nn.Module):
    _narrow = None      # InnerCos2 narrows the channel dim to 512

    def __init__(self, crit='MSE', strength=1, skip=0):
        super(InnerCos, self).__init__()
        self.crit = crit
        if crit != 'MSE':
            raise NotImplementedError("InnerCos: only the MSE criterion (the reference default, models/networks.py:312) "
                                      "has a HIP kernel")
        self.criterion = torch.nn.MSELoss()     # kept for attribute parity; the fused kernel computes the value
        self.strength = strength
        self.target = None
        self.skip = skip
        self.mask = None
        self.loss = 0

    def set_mask(self, mask_global, opt, feat_mask=None):
        """reference InnerCos.py:16-21: the 3-level feature mask as a float [h,w] tensor."""
        if feat_mask is not None:
            mask = feat_mask
        elif mask_global.size(0) > 1:                                   # per-sample masks (extension, see IPSR_model)
            mask = util.cal_feat_mask_batch(mask_global, 3, opt.threshold)
        else:
            mask = util.cal_feat_mask(mask_global, 3, opt.threshold)
        self.mask = (mask[:, 0] if mask.size(0) > 1 else mask.squeeze()).float()

    def set_target(self, targetIn):
        self.target = targetIn

    def get_target(self):
        return self.target

    def forward(self, in_data):
        if not self.skip:
            cuse = in_data.size(1) if self._narrow is None else self._narrow
            self.bs, self.c = in_data.size(0), cuse
            self.former = in_data if self._narrow is None else in_data.narrow(1, 0, cuse)
            if self.mask.dim() == 3:      # per-sample masks: equal-sized samples, so the batch mean is the mean of the per-sample means
                per = [_InnerCosLoss.apply(in_data[b:b + 1], cuse, self.mask[b], self.target[b:b + 1], float(self.strength))
                       for b in range(in_data.size(0))]
                self.loss = torch.stack(per).mean()
            else:
                self.loss = _InnerCosLoss.apply(in_data, cuse, self.mask, self.target, float(self.strength))
            self.output = in_data
        else:
            self.loss = 0
            self.output = in_data
        return self.output

    def backward(self, retain_graph=True):
        if not self.skip:
            self.loss.backward(retain_graph=retain_graph)
        return self.loss

    def __repr__(self):
        skip_str = 'True' if not self.skip else 'False'
        return self.__class__.__name__ + '(' \
            + 'skip: ' + skip_str \
            + ' ,strength: ' + str(self.strength) + ')'
