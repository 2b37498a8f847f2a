"""IPSRFunction — the reference's autograd surface (models/IPSRFunction.py) on libipsr_hip.so.

    out = IPSRFunction.apply(input, mask, ref, shift_sz, stride, triple_w, flag, nonmask_point_idx,
                             mask_point_idx, flatten_offsets, sp_x, sp_y)

Same 12 arguments, same output [B,C,h,w], backward returns (grad_input, None x 11) exactly like the
reference (:178).  What differs is only HOW: the reference loops over positions in Python and
materialises the N x N attention matrix `kbar` per sample (:36,78,134); here one C-ABI call launches the
fused kernels (normalise -> MFMA correlation + arg-max -> recurrence -> reconstruction) and keeps `kbar`
in sparse form (the masked rows + a CSR of the one-hot rows).

shift_sz > 1 (BASELINE config 4's 3x3 patches): the reference's forward computes the result and then raises while
saving `kbar` into a mis-sized buffer (:134); here it works — forward as the reference computes it up to :133,
backward as the same rule carried through the unfold/fold pair (include/ipsr_hip.h, ipsr_backward_patch).
"""
import torch

from .. import ops


class IPSRFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, input, mask, ref, shift_sz, stride, triple_w, flag, nonmask_point_idx, mask_point_idx,
                flatten_offsets, sp_x, sp_y):
        assert input.dim() == 4, "Input Dim has to be 4"
        assert mask.dim() == 2, "Mask dimension must be 2"
        ctx.triple_w = triple_w
        ctx.flag = flag
        ctx.flatten_offsets = flatten_offsets
        ctx.bz, _, ctx.h, ctx.w = input.size()

        # the index tensors must belong to THIS geometry: they live on the window grid (h-p+1)(w-p+1)
        n_win = (ctx.h - int(shift_sz) + 1) * (ctx.w - int(shift_sz) + 1)
        if torch.is_tensor(flag) and flag.numel() != n_win:
            raise RuntimeError("IPSRFunction: flag has %d entries but a %dx%d feature with shift_sz=%d has %d patch positions"
                               % (flag.numel(), ctx.h, ctx.w, int(shift_sz), n_win))
        mpi32 = getattr(mask_point_idx, "_ipsr_i32", None)
        if mpi32 is None or mpi32.device != input.device:
            # a foreign index tensor (not produced by util.cal_mask_given_mask_thred here): check its range once
            mpi32 = mask_point_idx.to(device=input.device, dtype=torch.int32)
            if mpi32.numel() and (int(mpi32.min()) < 0 or int(mpi32.max()) >= n_win):
                raise RuntimeError("IPSRFunction: mask_point_idx out of range [0, %d)" % n_win)
        # `ref` is the VGG namedtuple; only relu4_3 is read (reference :49)
        need_grad = ctx.needs_input_grad[0]      # grad mode is off inside Function.forward, so ask the ctx
        # the layer itself is fp32 whatever the surrounding autocast regime (BASELINE config 5 runs the convs in bf16)
        ctx.in_dtype = input.dtype
        f = ops.forward(input.detach().float(), ref.relu4_3.detach().float(), mpi32, int(shift_sz), int(stride),
                        want_index=need_grad)
        ctx.M = int(mpi32.numel())
        ctx.shift_sz = int(shift_sz)
        ctx.bwd_index = f.bwd_index        # sparse trunc(kbar)  (the reference keeps the dense ctx.ind_lst, :139)
        ctx.ind = f.ind
        ctx.vmax = f.vmax
        return f.out

    @staticmethod
    def backward(ctx, grad_output):
        grad_input = ops.backward(grad_output.float(), ctx.bwd_index, ctx.triple_w, ctx.M, ctx.shift_sz).to(ctx.in_dtype)
        return grad_input, None, None, None, None, None, None, None, None, None, None, None


# Test hook: callable(bwd_index [B, ints] int32 device tensor) -> the tensor the backward should use instead.  The backward's
# only input besides the gradient is this sparse trunc(kbar); a test that wants two runs to share ONE truncation (DESIGN.md §6:
# an ulp in a 512-long dot product switches a whole gradient column on or off) records it in the first run and replays it in
# the second, or replays the reference's own (tests/golden/trainer_step.npz).  None (always, outside those tests) = untouched.
bwd_index_hook = None


class IPSRFunctionDeviceCounts(torch.autograd.Function):
    """The same layer with the masked positions described ON THE DEVICE: `mpi32` [Mcap] (one index for the batch) or
    [B,Mcap] (one row per sample) and `counts` [B] int32 — what IPSR_model uses internally (include/ipsr_hip.h,
    ipsr_forward_masks).  Not part of the reference's surface (its IPSRFunction.apply above stays the foreign-caller entry);
    it exists because the reference's 12 host-sized index tensors force a device->host round trip per mask, and because a
    batch with one hole per sample cannot be described by them at all."""

    @staticmethod
    def forward(ctx, input, ref_feat, mpi32, counts, shift_sz, stride, triple_w):
        assert input.dim() == 4, "Input Dim has to be 4"
        ctx.in_dtype = input.dtype
        f = ops.forward(input.detach().float(), ref_feat.detach().float(), mpi32, int(shift_sz), int(stride),
                        want_index=ctx.needs_input_grad[0], counts=counts)
        ctx.Mcap, ctx.shift_sz, ctx.triple_w = int(mpi32.size(-1)), int(shift_sz), triple_w
        ctx.bwd_index, ctx.ind, ctx.vmax = f.bwd_index, f.ind, f.vmax
        if bwd_index_hook is not None and f.bwd_index is not None:
            ctx.bwd_index = bwd_index_hook(f.bwd_index)
        return f.out

    @staticmethod
    def backward(ctx, grad_output):
        g = ops.backward(grad_output.float(), ctx.bwd_index, ctx.triple_w, ctx.Mcap, ctx.shift_sz).to(ctx.in_dtype)
        return g, None, None, None, None, None, None
