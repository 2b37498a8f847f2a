"""IPSR_model — stateful nn.Module wrapper of the patch-attention layer (reference models/IPSR_model.py).

Same constructor, `set_mask`, `set_ref`, `forward` and repr.  The reference recomputes the index
tensors with an O(N^2) Python loop on EVERY forward because `cal_fixed_flag` is never cleared
(:45-53); they depend on the mask only, so here they are recomputed only after `set_mask`.
"""
import torch
import torch.nn as nn

from ..util import util
from .IPSRFunction import IPSRFunction


class IPSR_model(nn.Module):
    def __init__(self, threshold, fixed_mask, shift_sz=1, stride=1, mask_thred=1, triple_weight=1):
        super(IPSR_model, self).__init__()
        self.threshold = threshold
        self.fixed_mask = fixed_mask
        self.shift_sz = shift_sz
        self.stride = stride
        self.mask_thred = mask_thred
        self.triple_weight = triple_weight
        self.cal_fixed_flag = True      # reference name; True = index tensors must be (re)computed
        self.sp_x = None
        self.sp_y = None
        self.mask = None
        self.ref = None
        self._index_shape = None

    def set_mask(self, mask_global, layer_to_last, threshold, feat_mask=None):
        """reference :30-33.  `feat_mask` (optional, [1,1,h,w] byte) lets the trainer share ONE
        cal_feat_mask result between this layer and the two InnerCos modules (the reference computes
        the same pyramid three times per set_input, models/IPSR.py:155-158)."""
        mask = feat_mask if feat_mask is not None else util.cal_feat_mask(mask_global, layer_to_last, threshold)
        self.mask = mask.squeeze()
        self.cal_fixed_flag = True
        return self.mask

    def set_ref(self, latent_ref):
        self.ref = latent_ref

    def forward(self, input):
        _, self.c, self.h, self.w = input.size()
        if self.cal_fixed_flag or self._index_shape != (self.h, self.w):
            latter = input.narrow(0, 0, 1).detach()
            self.flag, self.nonmask_point_idx, self.flatten_offsets, self.mask_point_idx = \
                util.cal_mask_given_mask_thred(latter.squeeze(0), self.mask, self.shift_sz, self.stride, self.mask_thred)
            self.cal_fixed_flag = False
            self._index_shape = (self.h, self.w)
        else:
            assert torch.is_tensor(self.flag), 'flag must have been figured out and has to be a tensor!'
        if not (torch.is_tensor(self.sp_x) or torch.is_tensor(self.sp_y)):
            self.sp_x, self.sp_y = util.cal_sps_for_Advanced_Indexing(self.h, self.w)
        return IPSRFunction.apply(input, self.mask, self.ref, self.shift_sz, self.stride, self.triple_weight,
                                  self.flag, self.nonmask_point_idx, self.mask_point_idx, self.flatten_offsets,
                                  self.sp_x, self.sp_y)

    def __repr__(self):
        return self.__class__.__name__ + '(' \
            + 'threshold: ' + str(self.threshold) \
            + ' ,triple_weight ' + str(self.triple_weight) + ')'
