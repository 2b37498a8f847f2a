"""ctypes front-end of the CPU oracle (oracle/ipsr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under deepinpainting_amd/ imports this module; the product path has no CPU fallback.

Every function takes/returns numpy arrays and mirrors one `*_cpu` entry point, which in turn mirrors
the HIP C-ABI of include/ipsr_hip.h.  Reference citations live in the C file.
"""
import ctypes
import os
import subprocess
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libipsr_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (a few seconds).  Called by __graft_entry__.build()."""
    src = os.path.join(_HERE, "ipsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libipsr_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed with status %d" % (what, rc))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def feat_mask_out_dim(n, layers=3):
    for _ in range(layers):
        n = (n + 2 - 4) // 2 + 1
    return n


def feat_mask(mask, layers=3, threshold=5 / 16.0):
    """util.cal_feat_mask.  mask: [H,W] (any 0/nonzero dtype) -> [h,w] uint8."""
    m = np.ascontiguousarray(mask != 0, dtype=np.uint8)
    H, W = m.shape
    out = np.zeros((feat_mask_out_dim(H, layers), feat_mask_out_dim(W, layers)), np.uint8)
    _chk(lib().ipsr_feat_mask_cpu(_p(m, ctypes.c_uint8), H, W, layers, ctypes.c_float(threshold),
                                  _p(out, ctypes.c_uint8)), "feat_mask")
    return out


IndexPrep = namedtuple("IndexPrep", ["flag", "nonmask_point_idx", "flatten_offsets", "mask_point_idx"])


def flatten_offsets_from_flag(flag):
    """The (dead) `flatten_offsets` of util.cal_mask_given_mask_thred (util/util.py:149-157):
    for i ascending, fo[i - m_i] = m_i with m_i = #masked positions before i (last write wins)."""
    flag = np.asarray(flag, np.int64)
    n = flag.shape[0]
    m = np.cumsum(flag) - flag
    fo = np.zeros(n, np.int64)
    fo[np.arange(n) - m] = m  # numpy assigns repeated indices in order: last one wins
    return fo


def index_prep(feat, patch=1, stride=1, mask_thred=1):
    """util.cal_mask_given_mask_thred -> (flag, nonmask_point_idx, flatten_offsets, mask_point_idx)."""
    f = np.ascontiguousarray(feat, dtype=np.uint8)
    h, w = f.shape
    n = ((h - patch) // stride + 1) * ((w - patch) // stride + 1)
    flag = np.zeros(n, np.int32)
    mpi = np.zeros(n, np.int32)
    cnt = np.zeros(1, np.int32)
    _chk(lib().ipsr_index_prep_cpu(_p(f, ctypes.c_uint8), h, w, patch, stride, mask_thred,
                                   _p(flag, ctypes.c_int32), _p(mpi, ctypes.c_int32), _p(cnt, ctypes.c_int32)),
         "index_prep")
    m = int(cnt[0])
    return IndexPrep(flag.astype(np.int64), np.arange(n, dtype=np.int64), flatten_offsets_from_flag(flag),
                     mpi[:m].astype(np.int64))


def patch_normalize(x):
    """x [B,C,N] -> (xn [B,C,N], inv [B,N])."""
    x = _f32(x)
    B, C, N = x.shape
    xn = np.empty_like(x)
    inv = np.empty((B, N), np.float32)
    _chk(lib().ipsr_patch_normalize_cpu(_p(x, ctypes.c_float), B, C, N, _p(xn, ctypes.c_float),
                                        _p(inv, ctypes.c_float)), "patch_normalize")
    return xn, inv


def corr_argmax(xn, ref, want_S=False):
    """xn, ref [B,C,N] -> (ind [B,N] i32, vmax [B,N], S [B,N,N] or None)."""
    xn, ref = _f32(xn), _f32(ref)
    B, C, N = xn.shape
    ind = np.empty((B, N), np.int32)
    vmax = np.empty((B, N), np.float32)
    S = np.empty((B, N, N), np.float32) if want_S else None
    _chk(lib().ipsr_corr_argmax_cpu(_p(xn, ctypes.c_float), _p(ref, ctypes.c_float), B, C, N,
                                    _p(ind, ctypes.c_int32), _p(vmax, ctypes.c_float), _p(S, ctypes.c_float)),
         "corr_argmax")
    return ind, vmax, S


def bwd_index_ints(N, M):
    lib().ipsr_bwd_index_ints_cpu.restype = ctypes.c_size_t
    return int(lib().ipsr_bwd_index_ints_cpu(N, M))


Forward = namedtuple("Forward", ["out", "ind", "vmax", "attn_rows", "bwd_index"])


def forward(x, ref, mask_point_idx, patch=1, stride=1):
    """IPSRFunction.forward.  x, ref [B,C,h,w]; mask_point_idx [M] -> Forward(...)."""
    x, ref = _f32(x), _f32(ref)
    B, C, h, w = x.shape
    N = (h - patch + 1) * (w - patch + 1)          # the window grid: ind / vmax / attn_rows / bwd_index live on it
    mpi = np.ascontiguousarray(mask_point_idx, dtype=np.int32)
    M = int(mpi.shape[0])
    out = np.empty_like(x)
    ind = np.empty((B, N), np.int32)
    vmax = np.empty((B, N), np.float32)
    attn = np.zeros((B, max(M, 1), N), np.float32)
    bidx = np.zeros((B, bwd_index_ints(N, M)), np.int32)
    _chk(lib().ipsr_forward_cpu(_p(x, ctypes.c_float), _p(ref, ctypes.c_float), _p(mpi, ctypes.c_int32), M,
                                B, C, h, w, patch, stride, _p(out, ctypes.c_float), _p(ind, ctypes.c_int32),
                                _p(vmax, ctypes.c_float), _p(attn, ctypes.c_float), _p(bidx, ctypes.c_int32)),
         "forward")
    return Forward(out, ind, vmax, attn[:, :M], bidx)


def unfold(x, patch):
    """x [B,C,h,w] -> [B, C*p*p, N'] (rows k = (c*p+dy)*p+dx)."""
    x = _f32(x)
    B, C, h, w = x.shape
    xu = np.empty((B, C * patch * patch, (h - patch + 1) * (w - patch + 1)), np.float32)
    _chk(lib().ipsr_unfold_cpu(_p(x, ctypes.c_float), B, C, h, w, patch, _p(xu, ctypes.c_float)), "unfold")
    return xu


def fold(yu, C, h, w, patch):
    """overlap-add of [B, C*p*p, N'] back to [B,C,h,w]."""
    yu = _f32(yu)
    B = yu.shape[0]
    out = np.empty((B, C, h, w), np.float32)
    _chk(lib().ipsr_fold_cpu(_p(yu, ctypes.c_float), B, C, h, w, patch, _p(out, ctypes.c_float)), "fold")
    return out


def backward(grad_out, mask_point_idx, attn_rows, bwd_index, triple_w=1.0):
    """IPSRFunction.backward.  grad_out [B,C,h,w] -> grad_in [B,C,h,w]."""
    g = _f32(grad_out)
    B, C, h, w = g.shape
    mpi = np.ascontiguousarray(mask_point_idx, dtype=np.int32)
    M = int(mpi.shape[0])
    attn = _f32(attn_rows) if M > 0 else np.zeros((B, 1, h * w), np.float32)
    bidx = np.ascontiguousarray(bwd_index, dtype=np.int32)
    gin = np.empty_like(g)
    _chk(lib().ipsr_backward_cpu(_p(g, ctypes.c_float), _p(mpi, ctypes.c_int32), M, _p(attn, ctypes.c_float),
                                 _p(bidx, ctypes.c_int32), ctypes.c_float(triple_w), B, C, h, w,
                                 _p(gin, ctypes.c_float)), "backward")
    return gin


def backward_patch(grad_out, M, bwd_index, triple_w=1.0, patch=1):
    """shift_sz > 1 extension of the backward (see ipsr_backward_patch_cpu)."""
    g = _f32(grad_out)
    B, C, h, w = g.shape
    bidx = np.ascontiguousarray(bwd_index, dtype=np.int32)
    gin = np.empty_like(g)
    _chk(lib().ipsr_backward_patch_cpu(_p(g, ctypes.c_float), int(M), _p(bidx, ctypes.c_int32), ctypes.c_float(triple_w),
                                       B, C, h, w, patch, _p(gin, ctypes.c_float)), "backward_patch")
    return gin


def innercos_loss(x, mask, target, strength=1.0):
    """InnerCos/InnerCos2 loss.  x [B,Cx,h,w], mask [h,w], target [B,Cuse,h,w] -> python float (fp32 value)."""
    x, target = _f32(x), _f32(target)
    B, Cx = x.shape[:2]
    Cuse = target.shape[1]
    N = int(np.prod(x.shape[2:]))
    m = _f32(mask).reshape(-1)
    loss = np.zeros(1, np.float32)
    _chk(lib().innercos_loss_cpu(_p(x, ctypes.c_float), B, Cx, Cuse, N, _p(m, ctypes.c_float),
                                 _p(target, ctypes.c_float), ctypes.c_float(strength), _p(loss, ctypes.c_float)),
         "innercos_loss")
    return loss[0]


def innercos_loss_backward(x, mask, target, strength=1.0, grad_loss=1.0):
    x, target = _f32(x), _f32(target)
    B, Cx = x.shape[:2]
    Cuse = target.shape[1]
    N = int(np.prod(x.shape[2:]))
    m = _f32(mask).reshape(-1)
    gl = np.array([grad_loss], np.float32)
    gx = np.empty_like(x)
    _chk(lib().innercos_loss_backward_cpu(_p(x, ctypes.c_float), B, Cx, Cuse, N, _p(m, ctypes.c_float),
                                          _p(target, ctypes.c_float), ctypes.c_float(strength),
                                          _p(gl, ctypes.c_float), _p(gx, ctypes.c_float)), "innercos_loss_backward")
    return gx
