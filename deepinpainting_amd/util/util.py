"""Mask / index helpers of the IPSR layer — the host-side mirror of the reference's util/util.py
(the hot-path part: :68-174).  Same names, same argument meaning, same assertion messages; the work is
done by the K1/K2 HIP kernels of libipsr_hip.so (include/ipsr_hip.h).
"""
import torch

from .. import ops


def cal_feat_mask(inMask, conv_layers, threshold):
    """reference util/util.py:68-84 — `conv_layers` x (4x4, stride 2, pad 1, weight 1/16) box convolutions
    of the 0/1 mask, thresholded ONCE after the last one.  inMask: [1,1,H,W] (bool / 0-1 valued).
    Returns a ByteTensor [1,1,H/2^L,W/2^L] on inMask's device."""
    assert inMask.dim() == 4, "mask must be 4 dimensions"
    assert inMask.size(0) == 1, "the first dimension must be 1 for mask"
    m = inMask[0, 0]
    if m.dtype not in (torch.bool, torch.uint8):
        m = m != 0          # the reference feeds a BoolTensor cast to float (models/IPSR.py:36); 0/1 only
    feat = ops.feat_mask(m, int(conv_layers), float(threshold))
    return feat[None, None]


def flatten_offsets_from_flag(flag):
    """The reference's `flatten_offsets` (util/util.py:149-157).  Dead data — IPSRFunction stores it on ctx
    and never reads it (models/IPSRFunction.py:21,88-89) — kept because it is part of the 12-argument
    autograd surface.  fo[i - m_i] = m_i for i ascending (last write wins), m_i = #masked before i."""
    flag = flag.to(torch.int64)
    n = flag.numel()
    m = torch.cumsum(flag, 0) - flag
    pos = torch.arange(n, device=flag.device) - m          # non-decreasing in i
    # last writer of every position = the largest i mapping to it
    last_i = torch.full((n,), -1, dtype=torch.int64, device=flag.device)
    last_i.scatter_reduce_(0, pos, torch.arange(n, device=flag.device), reduce="amax", include_self=True)
    fo = torch.zeros(n, dtype=torch.int64, device=flag.device)
    valid = last_i >= 0
    fo[valid] = m[last_i[valid]]
    return fo


def cal_mask_given_mask_thred(img, mask, patch_size, stride, mask_thred):
    """reference util/util.py:88-161.  img [C,h,w] (shape only), mask [h,w] byte.
    Returns (flag [N], nonmask_point_idx [N] = arange(N), flatten_offsets [N], mask_point_idx [M]) as
    LongTensors on the mask's device.  One host sync (the count M)."""
    assert img.dim() == 3, 'img has to be 3 dimenison!'
    assert mask.dim() == 2, 'mask has to be 2 dimenison!'
    m = mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)
    flag32, mpi32, cnt = ops.index_prep(m, int(patch_size), int(stride), int(mask_thred))
    M = int(cnt.item())
    flag = flag32.to(torch.int64)
    mask_point_idx = mpi32[:M].to(torch.int64)
    mask_point_idx._ipsr_i32 = mpi32[:M].contiguous()      # what the kernels consume; saves a cast per forward
    nonmask_point_idx = torch.arange(flag.numel(), dtype=torch.int64, device=flag.device)
    return flag, nonmask_point_idx, flatten_offsets_from_flag(flag), mask_point_idx


def cal_sps_for_Advanced_Indexing(h, w):
    """reference util/util.py:166-174 (dead in the layer: MaxCoord ignores them, util/MaxCoord.py:25)."""
    sp_y = torch.arange(0, w).long().repeat(h)
    sp_x = torch.arange(0, h).long().repeat_interleave(w)
    return sp_x, sp_y
