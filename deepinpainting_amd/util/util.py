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


def cal_feat_mask_batch(inMask, conv_layers, threshold):
    """Per-sample masks (extension): cal_feat_mask of every [1,1,H,W] slice of a [B,1,H,W] mask -> [B,1,h,w] byte."""
    assert inMask.dim() == 4, "mask must be 4 dimensions"
    return torch.cat([cal_feat_mask(inMask[b:b + 1], conv_layers, threshold) for b in range(inMask.size(0))], 0)


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


# ----------------------------------------------------------------------------------------------------
# Image / bookkeeping helpers of the reference's util/util.py that its drivers import next to the mask
# functions (train.ipynb, test.ipynb, app.py).  Host-side, no GPU work; present so that the package alias
# of INTEGRATION.md §2a covers the whole `util.util` surface.
# ----------------------------------------------------------------------------------------------------
def tensor2im(image_tensor, imtype=None):
    """reference :15-21 — first image of a [-1,1] batch as an HxWx3 array scaled to [0,255] (grey -> 3 channels)."""
    import numpy as np
    arr = image_tensor[0].detach().float().cpu().numpy()
    if arr.shape[0] == 1:
        arr = np.repeat(arr, 3, axis=0)
    arr = (arr.transpose(1, 2, 0) + 1.0) * 127.5
    return arr.astype(np.uint8 if imtype is None else imtype)


def diagnose_network(net, name='network'):
    """reference :23-31 — mean absolute gradient over the parameters that have one (the reference computes it and
    drops it; here it is returned as well)."""
    vals = [p.grad.detach().abs().mean() for p in net.parameters() if p.grad is not None]
    return float(torch.stack(vals).mean()) if vals else 0.0


def binary_mask(in_mask, threshold):
    """reference :33-39.  NB the reference thresholds an UNINITIALISED ByteTensor of the mask's size (:36-37), i.e. its
    result does not depend on `in_mask`; no caller uses it.  This one thresholds the mask itself."""
    assert in_mask.dim() == 2, "mask must be 2 dimensions"
    return (in_mask > threshold).float()


def create_gMask(gMask_opts):
    """reference :41-64 — crop a fineSize window whose masked area is between 20 % and maxPartition % out of a big
    pattern.  `pattern` [MAX_SIZE,MAX_SIZE] 0/1 tensor; returns it expanded to mask_global's rank."""
    import random
    pattern = gMask_opts['pattern']
    mask_global = gMask_opts['mask_global']
    big, fine, most = gMask_opts['MAX_SIZE'], gMask_opts['fineSize'], gMask_opts['maxPartition']
    if pattern is None:
        raise ValueError
    while True:
        x, y = random.randint(1, big - fine), random.randint(1, big - fine)
        window = pattern[y:y + fine, x:x + fine]
        percent = float(window.sum()) * 100.0 / (fine * fine)
        if 20 < percent < most:
            break
    lead = (1,) if mask_global.dim() == 3 else (1, 1)
    return window.expand(*lead, window.size(0), window.size(1))


def save_image(image_numpy, image_path):
    """reference :177-179."""
    from PIL import Image
    Image.fromarray(image_numpy).save(image_path)


def info(object, spacing=10, collapse=1):
    """reference :181-191 — print the callables of an object with their doc strings."""
    squeeze = (lambda s: " ".join(s.split())) if collapse else (lambda s: s)
    for name in dir(object):
        attr = getattr(object, name)
        if callable(attr):
            print("%s %s" % (name.ljust(spacing), squeeze(str(attr.__doc__))))
