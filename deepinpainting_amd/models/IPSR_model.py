"""IPSR_model — stateful nn.Module wrapper of the patch-attention layer (reference models/IPSR_model.py).

Same constructor, `set_mask`, `set_ref`, `forward` and repr.  The reference recomputes the index
tensors with an O(N^2) Python loop on EVERY forward because `cal_fixed_flag` is never cleared
(:45-53); they depend on the mask only, so here they are recomputed only after `set_mask`.

Extension (not in the reference, whose notebook pins batchSize = 1 because it has ONE mask per batch): `set_mask` with a
[B,1,H,W] mask gives every sample its own hole.  The layer then runs sample by sample — each sample has its own masked
positions and its own recurrence length — which is exactly the reference's per-sample loop (models/IPSRFunction.py:46) with
the mask inside it; per-sample results equal a batch-of-one call with that sample's mask.
"""
import torch
import torch.nn as nn

from ..util import util
from .IPSRFunction import IPSRFunction


class _Ref(object):
    """Minimal stand-in for the VGG namedtuple: IPSRFunction reads `.relu4_3` only (reference :49)."""

    def __init__(self, relu4_3):
        self.relu4_3 = relu4_3


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
        self.corr_bf16 = False          # opt-in (BASELINE config 5): correlation on the bf16 MFMA kernel; fp32 = the reference
        self.sp_x = None
        self.sp_y = None
        self.mask = None
        self.ref = None
        self._index_shape = None

    def set_mask(self, mask_global, layer_to_last, threshold, feat_mask=None):
        """reference :30-33.  `feat_mask` (optional, [1,1,h,w] byte) lets the trainer share ONE
        cal_feat_mask result between this layer and the two InnerCos modules (the reference computes
        the same pyramid three times per set_input, models/IPSR.py:155-158)."""
        if feat_mask is not None:
            mask = feat_mask
        elif mask_global.size(0) > 1:
            mask = util.cal_feat_mask_batch(mask_global, layer_to_last, threshold)
        else:
            mask = util.cal_feat_mask(mask_global, layer_to_last, threshold)
        self.mask = mask[:, 0] if mask.size(0) > 1 else mask.squeeze()      # [h,w], or [B,h,w] with per-sample masks
        self.cal_fixed_flag = True
        return self.mask

    def set_ref(self, latent_ref):
        self.ref = latent_ref

    def _forward_per_sample(self, input):
        B = input.size(0)
        assert self.mask.size(0) == B, 'per-sample masks: %d masks for a batch of %d' % (self.mask.size(0), B)
        if self.cal_fixed_flag or self._index_shape != (self.h, self.w):
            self._per_sample_index = [util.cal_mask_given_mask_thred(input[0].detach(), self.mask[b], self.shift_sz, self.stride,
                                                                     self.mask_thred) for b in range(B)]
            self.cal_fixed_flag = False
            self._index_shape = (self.h, self.w)
        if not (torch.is_tensor(self.sp_x) or torch.is_tensor(self.sp_y)):
            self.sp_x, self.sp_y = util.cal_sps_for_Advanced_Indexing(self.h, self.w)
        feat = self.ref.relu4_3

        def one(b):
            flag, nonmask, offsets, mpi = self._per_sample_index[b]
            ref_b = self.ref._replace(relu4_3=feat[b:b + 1]) if hasattr(self.ref, '_replace') else _Ref(feat[b:b + 1])
            return IPSRFunction.apply(input[b:b + 1], self.mask[b], ref_b, self.shift_sz, self.stride, self.triple_weight,
                                      flag, nonmask, mpi, offsets, self.sp_x, self.sp_y)

        if not input.is_cuda or B == 1:
            return torch.cat([one(b) for b in range(B)], 0)
        # A single sample fills an eighth of the chip (64 correlation workgroups, one recurrence wave): run the samples
        # side by side on their own HIP streams.  Autograd replays each sample's backward on the stream of its forward.
        cur = torch.cuda.current_stream(input.device)
        if getattr(self, '_streams', None) is None or len(self._streams) < B or self._streams[0].device != input.device:
            self._streams = [torch.cuda.Stream(device=input.device) for _ in range(B)]
        outs = [None] * B
        for b in range(B):
            st = self._streams[b]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs[b] = one(b)
        for b in range(B):
            cur.wait_stream(self._streams[b])
            outs[b].record_stream(cur)
        return torch.cat(outs, 0)

    def forward(self, input):
        if self.corr_bf16 and input.is_cuda:
            from .. import ops
            with ops.corr_precision("bf16"):
                return self._forward(input)
        return self._forward(input)

    def _forward(self, input):
        _, self.c, self.h, self.w = input.size()
        if self.mask is not None and self.mask.dim() == 3:
            return self._forward_per_sample(input)
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
