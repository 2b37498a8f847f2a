"""IPSR_model — stateful nn.Module wrapper of the patch-attention layer (reference models/IPSR_model.py).

Same constructor, `set_mask`, `set_ref`, `forward` and repr.  The reference recomputes the index
tensors with an O(N^2) Python loop on EVERY forward because `cal_fixed_flag` is never cleared
(:45-53); they depend on the mask only, so here they are recomputed only after `set_mask`.

The index (flag, mask_point_idx, count) stays ON THE DEVICE: the K2 kernel's outputs go straight into the layer
(ipsr_forward_masks reads the count from device memory), so a forward costs no host round trip; the reference-surface
attributes `flag`, `nonmask_point_idx`, `flatten_offsets`, `mask_point_idx` are materialised on first access only.

Extension (not in the reference, whose notebook pins batchSize = 1 because it has ONE mask per batch): `set_mask` with a
[B,1,H,W] mask gives every sample its own hole.  The batch still runs as ONE launch sequence — every sample with its own
index row, count and recurrence length — which is exactly the reference's per-sample loop (models/IPSRFunction.py:46) with
the mask inside it; per-sample results equal a batch-of-one call with that sample's mask bit for bit.
"""
import torch
import torch.nn as nn

from ..util import util
from .IPSRFunction import IPSRFunction, IPSRFunctionDeviceCounts


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
        self._flag32 = self._mpi32 = self._counts = None
        self._host_index = None
        self._mask_src = None           # (tensor, version, layer_to_last, threshold) of the last set_mask: an unchanged mask keeps its index
        # capacity of the device-side index (columns of mask_point_idx handed to the kernels; everything the layer sizes by M —
        # compressed attention, backward CSR, LDS of the compress kernel — is sized by it):
        #   "auto"  the largest per-sample count, rounded up to 32: ONE host read for the FIRST mask (set_mask with the same, unmodified
        #           mask tensor keeps the index and costs nothing); from the second DIFFERENT mask on — a training loop that draws a
        #           mask per iteration, train.ipynb c2:16-19 — the capacity is N and nothing is read back: the read stalls the host
        #           behind the whole queued step (measured: 331 vs 340 images/s), sizing the buffers by N costs nothing measurable;
        #   "full"  N, never a host read;   int  caller-supplied bound (checked against the mask: one host read per new mask)
        self.index_capacity = "auto"
        self._masks_seen = 0

    def set_mask(self, mask_global, layer_to_last, threshold, feat_mask=None):
        """reference :30-33.  `feat_mask` (optional, [1,1,h,w] byte) lets the trainer share ONE
        cal_feat_mask result between this layer and the two InnerCos modules (the reference computes
        the same pyramid three times per set_input, models/IPSR.py:155-158)."""
        src = self._mask_src
        if feat_mask is None and src is not None and src[0] is mask_global and src[1] == mask_global._version \
                and src[2:] == (layer_to_last, threshold) and self.mask is not None:
            return self.mask                 # same mask tensor, not modified since: the feature mask and the index stand
        if feat_mask is not None:
            mask = feat_mask
        elif mask_global.size(0) > 1:
            mask = util.cal_feat_mask_batch(mask_global, layer_to_last, threshold)
        else:
            mask = util.cal_feat_mask(mask_global, layer_to_last, threshold)
        self.mask = mask[:, 0] if mask.size(0) > 1 else mask.squeeze()      # [h,w], or [B,h,w] with per-sample masks
        self.cal_fixed_flag = True
        self._masks_seen += 1
        self._mask_src = (mask_global, mask_global._version, layer_to_last, threshold) if feat_mask is None else None
        return self.mask

    def set_ref(self, latent_ref):
        self.ref = latent_ref

    # ---- device-side index: flag / mask_point_idx / count straight from the K2 kernel, never read back in the hot path ----
    def _index_rows(self, masks):
        """masks [h,w] or [B,h,w] byte -> (flag32 [R,N], mpi32 [R,N] (-1 padded), counts [R] int32), R = 1 or B."""
        from .. import ops
        rows = [masks] if masks.dim() == 2 else list(masks)
        outs = [ops.index_prep(m if m.dtype == torch.uint8 else m.to(torch.uint8), int(self.shift_sz), int(self.stride), int(self.mask_thred))
                for m in rows]
        return (torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), torch.cat([o[2] for o in outs]))

    def _ensure_index(self, input):
        # the device index and the reference-surface (CPU) index are cached separately: a CPU forward must not pass for a valid
        # device index at the same (h, w), nor an index built on another device
        if self.cal_fixed_flag or self._index_shape != (self.h, self.w) or self._flag32 is None or self._flag32.device != input.device:
            assert self.mask is not None, 'set_mask() must be called before forward()'
            mask = self.mask if self.mask.device == input.device else self.mask.to(input.device)
            flag32, mpi32, counts = self._index_rows(mask)
            n_win = (self.h - int(self.shift_sz) + 1) * (self.w - int(self.shift_sz) + 1)
            assert flag32.size(1) == n_win, 'mask %s does not match a %dx%d feature' % (tuple(self.mask.shape), self.h, self.w)
            cap = self.index_capacity
            if cap == "auto" and self._masks_seen > 1:
                cap = "full"                        # masks change from step to step: no read-back (see __init__)
            if cap == "auto":
                cap = int(counts.max().item())      # one host read for the first mask, off the per-forward path
            elif cap == "full":
                cap = n_win
            else:
                # a caller-supplied bound is checked against the mask it is used with (the same one host read "auto" makes): the
                # kernels clamp every count to the capacity, so a bound below the true count would silently drop masked positions
                need = int(counts.max().item())
                if need > int(cap):
                    raise ValueError("IPSR_model.index_capacity = %d, but this mask has %d masked feature positions "
                                     "(use 'auto', 'full' or a larger bound)" % (int(cap), need))
            cap = min(n_win, max(32, (int(cap) + 31) // 32 * 32))
            self._flag32, self._counts = flag32, counts
            self._mpi32 = mpi32[:, :cap].contiguous()          # entries past counts[b] are never read; the kernels size by `cap`
            self._host_index = None                 # the reference-surface tensors are rebuilt lazily
            self.cal_fixed_flag = False
            self._index_shape = (self.h, self.w)

    def _host_tensors(self):
        """flag / nonmask_point_idx / flatten_offsets / mask_point_idx as the reference exposes them (int64, exact sizes): built on
        first access (one host sync), for tests and foreign callers — the layer itself never needs them."""
        if getattr(self, '_host_index', None) is None:
            assert getattr(self, '_flag32', None) is not None, 'flag must have been figured out and has to be a tensor!'
            rows = []
            for r in range(self._flag32.size(0)):
                M = int(self._counts[r].item())
                flag = self._flag32[r].to(torch.int64)
                mpi = self._mpi32[r, :M].to(torch.int64)
                mpi._ipsr_i32 = self._mpi32[r, :M].contiguous()
                rows.append((flag, torch.arange(flag.numel(), dtype=torch.int64, device=flag.device), util.flatten_offsets_from_flag(flag), mpi))
            self._host_index = rows
        return self._host_index

    flag = property(lambda self: self._host_tensors()[0][0])
    nonmask_point_idx = property(lambda self: self._host_tensors()[0][1])
    flatten_offsets = property(lambda self: self._host_tensors()[0][2])
    mask_point_idx = property(lambda self: self._host_tensors()[0][3])

    @property
    def _per_sample_index(self):
        return self._host_tensors()

    def forward(self, input):
        if self.corr_bf16 and input.is_cuda:
            from .. import ops
            with ops.corr_precision("bf16"):
                return self._forward(input)
        return self._forward(input)

    def _forward(self, input):
        B, self.c, self.h, self.w = input.size()
        if not input.is_cuda:
            return self._forward_reference_surface(input)
        self._ensure_index(input)
        if self.mask.dim() == 3:
            # one hole per sample (extension): ONE launch sequence for the batch, every sample with its own index row and count
            assert self.mask.size(0) == B, 'per-sample masks: %d masks for a batch of %d' % (self.mask.size(0), B)
            mpi, counts = self._mpi32, self._counts
        else:
            mpi, counts = self._mpi32[0], self._counts.expand(B).contiguous()
        return IPSRFunctionDeviceCounts.apply(input, self.ref.relu4_3, mpi, counts, self.shift_sz, self.stride, self.triple_weight)

    def _forward_reference_surface(self, input):
        """The reference's own call sequence (models/IPSR_model.py:45-63) through the 12-argument IPSRFunction.apply: what a CPU
        tensor gets (the oracle-backed twin patches IPSRFunction for that) — the GPU path above is the same computation."""
        if self.mask.dim() == 3:
            raise NotImplementedError("per-sample masks need the GPU path")
        if self.cal_fixed_flag or self._index_shape != (self.h, self.w) or self._flag32 is not None or self._host_index is None:
            latter = input.narrow(0, 0, 1).detach()
            self._host_index = [util.cal_mask_given_mask_thred(latter.squeeze(0), self.mask, self.shift_sz, self.stride, self.mask_thred)]
            self._flag32 = self._mpi32 = self._counts = None     # the device index (if any) is stale now: _ensure_index rebuilds it
            self.cal_fixed_flag = False
            self._index_shape = (self.h, self.w)
        if not (torch.is_tensor(self.sp_x) or torch.is_tensor(self.sp_y)):
            self.sp_x, self.sp_y = util.cal_sps_for_Advanced_Indexing(self.h, self.w)
        flag, nonmask, offsets, mpi = self._host_index[0]
        return IPSRFunction.apply(input, self.mask, self.ref, self.shift_sz, self.stride, self.triple_weight,
                                  flag, nonmask, mpi, offsets, self.sp_x, self.sp_y)

    def __repr__(self):
        return self.__class__.__name__ + '(' \
            + 'threshold: ' + str(self.threshold) \
            + ' ,triple_weight ' + str(self.triple_weight) + ')'
