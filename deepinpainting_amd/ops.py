"""Thin torch-tensor front-ends of the C-ABI entry points (device pointers + current stream).

torch is used here for device memory and streams only; every computation is a HIP kernel of
libipsr_hip.so.  All functions require CUDA(HIP) tensors and raise otherwise — no fallback.
"""
import contextlib
import threading
import weakref
from collections import namedtuple

import torch

from . import _lib

_ws_cache = {}


def _stream():
    """The raw hipStream_t of torch's current stream on the current device (the C getter: no Stream object per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _workspace(nbytes, device):
    """Grow-only per-(device, stream) scratch buffer; kernels on one stream run in order, so reuse is safe."""
    key = (device.index, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _on_current_device(t, name):
    # the kernels are launched on the CURRENT device's stream with raw pointers: a tensor living on another GPU would fault
    if t.device.index != torch._C._cuda_getDevice():        # (the C getter: this check runs ~700 times per training step)
        raise RuntimeError("%s is on %s but the current device is cuda:%d — wrap the call in torch.cuda.device(...) "
                           "(one process per GPU sets it once)" % (name, t.device, torch.cuda.current_device()))


def _req(t, dtype, name):
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor — the IPSR layer has no CPU path" % name)
    _on_current_device(t, name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def feat_mask_out_dim(n, layers=3):
    for _ in range(layers):
        n = (n + 2 - 4) // 2 + 1
    return n


def feat_mask(mask_hw, layers, threshold):
    """K1.  mask_hw: [H,W] bool/uint8 CUDA tensor -> [h,w] uint8."""
    m = mask_hw
    if m.dtype == torch.bool:
        m = m.to(torch.uint8)
    m = _req(m, torch.uint8, "mask")
    H, W = m.shape
    out = torch.empty((feat_mask_out_dim(H, layers), feat_mask_out_dim(W, layers)), dtype=torch.uint8, device=m.device)
    L = _lib.lib()
    nbytes = L.ipsr_feat_mask_workspace_bytes(H, W, layers)
    ws = _workspace(nbytes, m.device)
    _lib.check(L.ipsr_feat_mask(m.data_ptr(), H, W, layers, float(threshold), out.data_ptr(), ws.data_ptr(),
                                ws.numel(), _stream()), "ipsr_feat_mask")
    return out


def index_prep(feat_hw, patch, stride, mask_thred):
    """K2.  feat_hw [h,w] uint8 -> (flag [N] i32, mask_point_idx [N] i32 (first M valid), count [1] i32)."""
    f = _req(feat_hw, torch.uint8, "feature mask")
    h, w = f.shape
    n = ((h - patch) // stride + 1) * ((w - patch) // stride + 1)
    flag = torch.empty(n, dtype=torch.int32, device=f.device)
    mpi = torch.empty(n, dtype=torch.int32, device=f.device)
    cnt = torch.empty(1, dtype=torch.int32, device=f.device)
    _lib.check(_lib.lib().ipsr_index_prep(f.data_ptr(), h, w, patch, stride, int(mask_thred), flag.data_ptr(),
                                          mpi.data_ptr(), cnt.data_ptr(), _stream()), "ipsr_index_prep")
    return flag, mpi, cnt


def patch_normalize(x_bcn):
    """K3.  x [B,C,N] -> (xn [B,C,N], inv [B,N])."""
    x = _req(x_bcn, torch.float32, "x")
    B, C, N = x.shape
    xn = torch.empty_like(x)
    inv = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ipsr_patch_normalize(x.data_ptr(), B, C, N, xn.data_ptr(), inv.data_ptr(), _stream()),
               "ipsr_patch_normalize")
    return xn, inv


def corr_argmax(xn_bcn, ref_bcn, want_S=False, corr="fp32"):
    """K4+K5.  -> (ind [B,N] i32, vmax [B,N], S [B,N,N] or None).
    corr="bf16": the opt-in bf16-MFMA kernel (operands rounded to bf16, fp32 accumulate); S is not available there."""
    xn = _req(xn_bcn, torch.float32, "xn")
    ref = _req(ref_bcn, torch.float32, "ref")
    B, C, N = xn.shape
    ind = torch.empty((B, N), dtype=torch.int32, device=xn.device)
    vmax = torch.empty((B, N), dtype=torch.float32, device=xn.device)
    if corr == "bf16":
        if want_S:
            raise ValueError("corr_argmax: the bf16 kernel does not materialise S")
        L = _lib.lib()
        ws = _workspace(L.ipsr_corr_argmax_bf16_workspace_bytes(B, C, N), xn.device)
        _lib.check(L.ipsr_corr_argmax_bf16(xn.data_ptr(), ref.data_ptr(), B, C, N, ind.data_ptr(), vmax.data_ptr(),
                                           ws.data_ptr(), ws.numel(), _stream()), "ipsr_corr_argmax_bf16")
        return ind, vmax, None
    if corr != "fp32":
        raise ValueError("corr must be 'fp32' or 'bf16'")
    S = torch.empty((B, N, N), dtype=torch.float32, device=xn.device) if want_S else None
    L = _lib.lib()
    ws = _workspace(L.ipsr_corr_argmax_workspace_bytes(B, C, N), xn.device)
    _lib.check(L.ipsr_corr_argmax(xn.data_ptr(), ref.data_ptr(), B, C, N, ind.data_ptr(), vmax.data_ptr(),
                                  S.data_ptr() if want_S else None, ws.data_ptr(), ws.numel(), _stream()),
               "ipsr_corr_argmax")
    return ind, vmax, S


Forward = namedtuple("Forward", ["out", "ind", "vmax", "attn_rows", "bwd_index"])


_precision = threading.local()


@contextlib.contextmanager
def corr_precision(corr):
    """`with ops.corr_precision("bf16"):` — layer forwards issued by this thread inside the block run their correlation on
    the bf16 MFMA kernel (BASELINE config 5).  The autograd surface IPSRFunction.apply has the reference's fixed 12
    arguments, so the choice travels this way (IPSR_model sets it from its `corr_bf16` attribute)."""
    if corr not in ("fp32", "bf16"):
        raise ValueError("corr must be 'fp32' or 'bf16'")
    prev = getattr(_precision, "corr", "fp32")
    _precision.corr = corr
    try:
        yield
    finally:
        _precision.corr = prev


def forward(x, ref, mask_point_idx_i32, patch=1, stride=1, want_attn=False, want_index=True, corr=None, counts=None):
    """Whole layer forward.  x, ref [B,C,h,w] fp32; mask_point_idx_i32 [M] i32 -> Forward.
    want_attn:  also materialise the dense attention rows [B,M,N] (the reference's `in_attention`; tests and
                inspection only — the layer itself works on the compressed form);
    want_index: build the sparse trunc(kbar) the backward needs (skip under no_grad);
    corr:       "fp32" (the reference's arithmetic, default) or "bf16" (opt-in bf16-MFMA correlation); None = what the
                enclosing `corr_precision` block says;
    counts:     [B] int32 DEVICE tensor = masked positions per sample (ipsr_forward_masks).  mask_point_idx is then [Mcap]
                (one index shared by the batch, e.g. straight from `index_prep`, no host sync) or [B,Mcap] (one row per
                sample: per-sample masks); entries past counts[b] are ignored.  Everything sized by M (attn_rows, bwd_index)
                is sized by the capacity Mcap; pass M = Mcap to `backward`."""
    if corr is None:
        corr = getattr(_precision, "corr", "fp32")
    if corr not in ("fp32", "bf16"):
        raise ValueError("corr must be 'fp32' or 'bf16'")
    x = _req(x, torch.float32, "input")
    ref = _req(ref, torch.float32, "ref.relu4_3")
    B, C, h, w = x.shape
    if tuple(ref.shape) != (B, C, h, w):
        raise RuntimeError("ref.relu4_3 %s does not match input %s" % (tuple(ref.shape), tuple(x.shape)))
    if stride != 1:
        raise NotImplementedError("IPSR layer: stride=%d is not implemented (only stride 1)" % stride)
    if patch < 1 or h < patch or w < patch:
        raise RuntimeError("IPSR layer: shift_sz=%d does not fit a %dx%d feature" % (patch, h, w))
    N = (h - patch + 1) * (w - patch + 1)       # window grid; = h*w for the reference's shift_sz = 1
    mpi = _req(mask_point_idx_i32, torch.int32, "mask_point_idx")
    if counts is not None:
        counts = _req(counts, torch.int32, "counts")
        if counts.numel() != B or mpi.dim() not in (1, 2) or (mpi.dim() == 2 and mpi.size(0) != B):
            raise RuntimeError("forward: counts must be [B] and mask_point_idx [Mcap] or [B,Mcap] (got %s, %s)" % (tuple(counts.shape), tuple(mpi.shape)))
        M = int(mpi.size(-1))
        if M < 1 or M > N:
            raise RuntimeError("forward: index capacity %d outside [1, %d]" % (M, N))
    else:
        M = int(mpi.numel())
    L = _lib.lib()
    dev = x.device
    out = torch.empty_like(x)
    ind = torch.empty((B, N), dtype=torch.int32, device=dev)
    vmax = torch.empty((B, N), dtype=torch.float32, device=dev)
    attn = torch.empty((B, M, N), dtype=torch.float32, device=dev) if (want_attn and M > 0) else None
    bidx = torch.empty((B, L.ipsr_bwd_index_ints(N, M)), dtype=torch.int32, device=dev) if want_index else None
    bf = corr == "bf16"
    nbytes = (L.ipsr_forward_bf16corr_workspace_bytes if bf else L.ipsr_forward_workspace_bytes)(B, C, h, w, M, patch, stride)
    ws = _workspace(nbytes, dev)
    if counts is not None:
        _lib.check(L.ipsr_forward_masks(x.data_ptr(), ref.data_ptr(), mpi.data_ptr(), M if mpi.dim() == 2 else 0, counts.data_ptr(), M,
                                        B, C, h, w, patch, stride, out.data_ptr(), ind.data_ptr(), vmax.data_ptr(),
                                        attn.data_ptr() if attn is not None else None,
                                        bidx.data_ptr() if bidx is not None else None, ws.data_ptr(), ws.numel(), _stream(), int(bf)),
                   "ipsr_forward_masks")
        return Forward(out, ind, vmax, attn, bidx)
    _lib.check((L.ipsr_forward_bf16corr if bf else L.ipsr_forward)(
        x.data_ptr(), ref.data_ptr(), mpi.data_ptr() if M else None, M, B, C, h, w,
        patch, stride, out.data_ptr(), ind.data_ptr(), vmax.data_ptr(),
        attn.data_ptr() if attn is not None else None,
        bidx.data_ptr() if bidx is not None else None, ws.data_ptr(), ws.numel(), _stream()),
        "ipsr_forward_bf16corr" if bf else "ipsr_forward")
    if want_attn and attn is None:
        attn = torch.empty((B, 0, N), dtype=torch.float32, device=dev)
    return Forward(out, ind, vmax, attn, bidx)


def backward(grad_out, bwd_index, triple_w, M, patch=1):
    """grad_in = g + triple_w * trunc(kbar)^T-weighted g; needs only the sparse index built by forward.
    patch > 1: the extension described at ipsr_backward_patch (include/ipsr_hip.h)."""
    g = _req(grad_out, torch.float32, "grad_output")
    B, C, h, w = g.shape
    gin = torch.empty_like(g)
    L = _lib.lib()
    if patch == 1:
        _lib.check(L.ipsr_backward(g.data_ptr(), None, int(M), None, bwd_index.data_ptr(), float(triple_w),
                                   B, C, h, w, gin.data_ptr(), _stream()), "ipsr_backward")
    else:
        ws = _workspace(L.ipsr_backward_workspace_bytes(B, C, h, w, patch), g.device)
        _lib.check(L.ipsr_backward_patch(g.data_ptr(), int(M), bwd_index.data_ptr(), float(triple_w), B, C, h, w, int(patch),
                                         gin.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "ipsr_backward_patch")
    return gin


def _req_io(t, name):
    """Activation tensors of the glue kernels: contiguous fp32 or bf16 on the GPU -> (tensor, io_bf16 flag)."""
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor" % name)
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))
    _on_current_device(t, name)
    return (t if t.is_contiguous() else t.contiguous()), int(t.dtype == torch.bfloat16)


def _f32(t):
    """bias / gamma / beta are fp32 parameters."""
    return None if t is None else _req(t, torch.float32, "parameter")


def bias_act_(x, bias, act="relu", slope=0.2, relu_into=None, relu_at=0, tickets=None):
    """In place x[b,c,...] = act(x + bias[c]) on a contiguous fp32/bf16 [B,C,*] tensor.  act: none | relu | leaky.
    relu_into / relu_at: a contiguous [B,Ctot,H,W] tensor whose channels [relu_at, relu_at + C) also receive relu(x + bias) — the skip
    half of the child level's concatenated tensor.  tickets: an int32 [C] tensor this launch zeroes for `bias_act_backward`'s in-launch
    batch sum of the bias gradient (the arrival counters live in the caller's memory, one set per autograd node)."""
    if not x.is_contiguous():
        raise RuntimeError("bias_act_ works in place and needs a contiguous tensor")
    x, bf = _req_io(x, "x")
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // (B * C)
    if relu_into is None:
        _lib.check(_lib.lib().ipsr_bias_act(x.data_ptr(), _ptr(_f32(bias)), B, C, hw, ACT_CODE[act], float(slope), bf, _ptr(tickets), _stream()),
                   "ipsr_bias_act")
        return x
    _check_wide(relu_into, x, "bias_act_: `relu_into`")
    if relu_at < 0 or relu_at + C > relu_into.shape[1]:
        raise RuntimeError("bias_act_: channels [%d, %d) outside `relu_into` %s" % (relu_at, relu_at + C, tuple(relu_into.shape)))
    y2p, y2bs = _slot(relu_into, relu_at, hw)
    _lib.check(_lib.lib().ipsr_bias_act_skip(x.data_ptr(), _ptr(_f32(bias)), B, C, hw, ACT_CODE[act], float(slope), bf, y2p, y2bs, _ptr(tickets),
                                             _stream()), "ipsr_bias_act_skip")
    return x


ACT_CODE = {"none": 0, "relu": 1, "leaky": 2}
INSTNORM_MAX_PLANE = 65536       # planes above 16384 elements must be a multiple of 4 (csrc/instnorm.hip)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _slot(t, c0, hw):
    """(pointer to channel c0 of a contiguous [B,Ctot,H,W] tensor, its batch stride in elements)."""
    return t.data_ptr() + c0 * hw * t.element_size(), t.shape[1] * hw


def _check_wide(wide, x, what):
    if wide.dtype != x.dtype or not wide.is_contiguous() or wide.dim() != 4 or wide.shape[0] != x.shape[0] \
            or tuple(wide.shape[2:]) != tuple(x.shape[2:]) or wide.device != x.device:
        raise RuntimeError("%s %s does not extend %s along the channels" % (what, tuple(wide.shape), tuple(x.shape)))


def instnorm_act_forward(x, bias, gamma, beta, eps, act, slope, into=None, into_at=0, relu_into=None, relu_at=0):
    """y = act(InstanceNorm(x + bias[c]) * gamma[c] + beta[c]) -> (y, mean [B*C], rstd [B*C]); x contiguous fp32 / bf16 [B,C,H,W].
    into / into_at: a contiguous [B,Ctot,H,W] tensor whose channels [into_at, into_at + C) receive y (a skip concatenation written
    in place); the returned y is then `into` itself.  relu_into / relu_at: a second destination of the same kind that receives
    relu(normalised value) — the skip half of the CHILD level's concatenated tensor."""
    x, bf = _req_io(x, "x")
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // (B * C)
    # one allocation: mean [B*C] | rstd [B*C] | the C ticket words of the backward's in-launch batch sums, zeroed by this launch
    stats = torch.empty(2 * B * C + C, dtype=torch.float32, device=x.device)
    mean, rstd, tickets = stats[:B * C], stats[B * C:2 * B * C], stats[2 * B * C:]
    if into is None and relu_into is None:
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ipsr_instnorm_act_forward(x.data_ptr(), _ptr(_f32(bias)), _ptr(_f32(gamma)), _ptr(_f32(beta)), float(eps),
                                                        ACT_CODE[act], float(slope), B, C, hw, bf, y.data_ptr(), mean.data_ptr(),
                                                        rstd.data_ptr(), tickets.data_ptr(), _stream()), "ipsr_instnorm_act_forward")
        return y, mean, rstd
    if into is not None:
        _check_wide(into, x, "instnorm_act_forward: `into`")
        if into_at < 0 or into_at + C > into.shape[1]:
            raise RuntimeError("instnorm_act_forward: channels [%d, %d) outside `into` %s" % (into_at, into_at + C, tuple(into.shape)))
        y, (yp, ybs) = into, _slot(into, into_at, hw)
    else:
        y = torch.empty_like(x)
        yp, ybs = y.data_ptr(), C * hw
    y2p, y2bs = None, 0
    if relu_into is not None:
        _check_wide(relu_into, x, "instnorm_act_forward: `relu_into`")
        if relu_at < 0 or relu_at + C > relu_into.shape[1]:
            raise RuntimeError("instnorm_act_forward: channels [%d, %d) outside `relu_into` %s" % (relu_at, relu_at + C, tuple(relu_into.shape)))
        y2p, y2bs = _slot(relu_into, relu_at, hw)
    _lib.check(_lib.lib().ipsr_instnorm_act_forward_slice(x.data_ptr(), _ptr(_f32(bias)), _ptr(_f32(gamma)), _ptr(_f32(beta)), float(eps),
                                                          ACT_CODE[act], float(slope), B, C, hw, bf, yp, ybs, y2p, y2bs,
                                                          mean.data_ptr(), rstd.data_ptr(), tickets.data_ptr(), _stream()),
               "ipsr_instnorm_act_forward_slice")
    return y, mean, rstd


def _tickets_behind(mean, rstd, B, C):
    """The ticket words `instnorm_act_forward` allocated behind its statistics (and zeroed in its launch); statistics that came from
    somewhere else get fresh zeroed words."""
    n = B * C
    if mean.dtype == torch.float32 and mean.numel() == n and rstd.numel() == n and rstd.data_ptr() == mean.data_ptr() + 4 * n \
            and mean.untyped_storage().nbytes() - 4 * mean.storage_offset() >= 4 * (2 * n + C):
        return mean.data_ptr() + 8 * n, None
    t = torch.zeros(C, dtype=torch.int32, device=mean.device)
    return t.data_ptr(), t


def instnorm_act_backward(dy, y, x, bias, gamma, mean, rstd, act, slope, need_affine, need_bias, at=0, dy2=None, dy2_at=0):
    """-> (dx, dgamma [C] | None, dbeta [C] | None, dbias [C] | None).  dy and y may be WIDER than x along the channels (contiguous
    [B,Ctot,H,W]): their channels [at, at + C) are the operands (a skip concatenation and its gradient, read in place).  dy2 / dy2_at:
    the gradient of the relu'd second output of the forward (channels [dy2_at, dy2_at + C) of a wide tensor), added inside the kernel."""
    x, bf = _req_io(x, "x")
    B, C = x.shape[0], x.shape[1]
    hw = x.numel() // (B * C)
    dy, _ = _req_io(dy.to(x.dtype), "grad_output")
    dx = torch.empty_like(x)
    part = torch.empty((3, B, C), dtype=torch.float32, device=x.device)
    # the batch sums of the per-plane partials are written by the same launch (the last plane of each channel to finish)
    sums = torch.empty((3, C), dtype=torch.float32, device=x.device)
    tick, _keep = _tickets_behind(mean, rstd, B, C)
    L = _lib.lib()
    if dy.shape[1] == C and y.shape[1] == C and dy2 is None:
        _lib.check(L.ipsr_instnorm_act_backward(dy.data_ptr(), y.data_ptr(), x.data_ptr(), _ptr(_f32(bias)), _ptr(_f32(gamma)),
                                                mean.data_ptr(), rstd.data_ptr(), ACT_CODE[act], float(slope), B, C, hw, bf,
                                                dx.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr(),
                                                _ptr(sums), tick, _stream()), "ipsr_instnorm_act_backward")
    else:
        _check_wide(dy, x, "instnorm_act_backward: grad_output")
        _check_wide(y, x, "instnorm_act_backward: output")
        if at < 0 or at + C > dy.shape[1] or (y.shape[1] != C and y.shape[1] != dy.shape[1]):
            raise RuntimeError("instnorm_act_backward: channels [%d, %d) outside %s / %s" % (at, at + C, tuple(dy.shape), tuple(y.shape)))
        dyp, dybs = _slot(dy, at if dy.shape[1] != C else 0, hw)
        yp, ybs = _slot(y, at if y.shape[1] != C else 0, hw)
        d2p, d2bs = None, 0
        if dy2 is not None:
            dy2, _ = _req_io(dy2.to(x.dtype), "second grad_output")
            _check_wide(dy2, x, "instnorm_act_backward: second grad_output")
            if dy2_at < 0 or dy2_at + C > dy2.shape[1]:
                raise RuntimeError("instnorm_act_backward: channels [%d, %d) outside the second gradient %s" % (dy2_at, dy2_at + C, tuple(dy2.shape)))
            d2p, d2bs = _slot(dy2, dy2_at, hw)
        _lib.check(L.ipsr_instnorm_act_backward_slice(dyp, dybs, d2p, d2bs, yp, ybs, x.data_ptr(), _ptr(_f32(bias)),
                                                      _ptr(_f32(gamma)), mean.data_ptr(), rstd.data_ptr(), ACT_CODE[act], float(slope), B, C, hw, bf,
                                                      dx.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), part[2].data_ptr(),
                                                      _ptr(sums), tick, _stream()), "ipsr_instnorm_act_backward_slice")
    s = sums
    return dx, (s[0] if need_affine else None), (s[1] if need_affine else None), (s[2] if need_bias else None)


def bias_act_backward(dy, y, act, slope, need_bias, dy2=None, dy2_at=0, tickets=None):
    """dx = dy * act'(y) (+ dy2[:, dy2_at : dy2_at + C] * relu'(y): the gradient of bias_act_'s second output), dbias [C] | None.
    tickets: the int32 [C] tensor the forward `bias_act_` zeroed (None: fresh zeroed words are allocated here)."""
    dy, bf = _req_io(dy.to(y.dtype), "grad_output")
    B, C = y.shape[0], y.shape[1]
    hw = y.numel() // (B * C)
    dx = torch.empty_like(y)
    part = torch.empty((B, C), dtype=torch.float32, device=y.device) if need_bias else None
    sums = torch.empty(C, dtype=torch.float32, device=y.device) if need_bias else None
    if need_bias and tickets is None:
        tickets = torch.zeros(C, dtype=torch.int32, device=y.device)
    if dy2 is None:
        _lib.check(_lib.lib().ipsr_bias_act_backward(dy.data_ptr(), y.data_ptr(), ACT_CODE[act], float(slope), B, C, hw, bf, dx.data_ptr(),
                                                     _ptr(part), _ptr(sums), _ptr(tickets), _stream()), "ipsr_bias_act_backward")
    else:
        dy2, _ = _req_io(dy2.to(y.dtype), "second grad_output")
        _check_wide(dy2, y, "bias_act_backward: second grad_output")
        if dy2_at < 0 or dy2_at + C > dy2.shape[1]:
            raise RuntimeError("bias_act_backward: channels [%d, %d) outside the second gradient %s" % (dy2_at, dy2_at + C, tuple(dy2.shape)))
        d2p, d2bs = _slot(dy2, dy2_at, hw)
        _lib.check(_lib.lib().ipsr_bias_act_backward_skip(dy.data_ptr(), d2p, d2bs, y.data_ptr(), ACT_CODE[act], float(slope), B, C, hw, bf,
                                                          dx.data_ptr(), _ptr(part), _ptr(sums), _ptr(tickets), _stream()),
                   "ipsr_bias_act_backward_skip")
    return dx, (sums if need_bias else None)


def cat_relu_forward(y, x):
    """relu(torch.cat([y, x], 1)) in one pass; y [B,C1,H,W], x [B,C2,H,W] contiguous, same fp32/bf16 dtype."""
    y, bf = _req_io(y, "y")
    x, bf2 = _req_io(x, "x")
    if bf != bf2 or y.shape[0] != x.shape[0] or y.shape[2:] != x.shape[2:]:
        raise RuntimeError("cat_relu: mismatched operands %s %s / %s %s" % (tuple(y.shape), y.dtype, tuple(x.shape), x.dtype))
    B, C1, C2 = y.shape[0], y.shape[1], x.shape[1]
    hw = y.numel() // (B * C1)
    out = torch.empty((B, C1 + C2) + tuple(y.shape[2:]), dtype=y.dtype, device=y.device)
    _lib.check(_lib.lib().ipsr_cat_relu_forward(y.data_ptr(), x.data_ptr(), B, C1, C2, hw, bf, out.data_ptr(), _stream()),
               "ipsr_cat_relu_forward")
    return out


def cat_relu_skip_half_(out, x):
    """out[:, C1:] = relu(x) for a contiguous out [B,C1+C2,H,W] whose first C1 channels are already written (instnorm_act_forward
    with `into`): the skip half of relu(torch.cat([y, x], 1))."""
    x, bf = _req_io(x, "x")
    out, bf2 = _req_io(out, "out")
    B, C2 = x.shape[0], x.shape[1]
    C1 = out.shape[1] - C2
    if bf != bf2 or out.shape[0] != B or C1 < 1 or out.shape[2:] != x.shape[2:]:
        raise RuntimeError("cat_relu_skip_half_: mismatched operands %s / %s" % (tuple(out.shape), tuple(x.shape)))
    hw = x.numel() // (B * C2)
    _lib.check(_lib.lib().ipsr_cat_relu_forward(None, x.data_ptr(), B, C1, C2, hw, bf, out.data_ptr(), _stream()), "ipsr_cat_relu_forward")
    return out


def cat_relu_backward(grad_out, out, C1, skip_half_only=False):
    """-> (dy [B,C1,..] | None, dx [B,C2,..]): the ReLU mask from `out`, the channel slices as contiguous tensors; skip_half_only: dy is
    not produced (its consumer reads grad_out's slice in place)."""
    g, bf = _req_io(grad_out.to(out.dtype), "grad_output")
    B, C = out.shape[0], out.shape[1]
    C2 = C - C1
    hw = out.numel() // (B * C)
    dy = None if skip_half_only else torch.empty((B, C1) + tuple(out.shape[2:]), dtype=out.dtype, device=out.device)
    dx = torch.empty((B, C2) + tuple(out.shape[2:]), dtype=out.dtype, device=out.device)
    _lib.check(_lib.lib().ipsr_cat_relu_backward(g.data_ptr(), out.data_ptr(), B, C1, C2, hw, bf, _ptr(dy), dx.data_ptr(), _stream()),
               "ipsr_cat_relu_backward")
    return dy, dx


def bias_relu_pool2(x, bias):
    """max_pool2d(relu(x + bias[c]), 2, 2) of a contiguous fp32 [B,C,H,W] tensor in one pass."""
    x, bf = _req_io(x, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().ipsr_bias_relu_pool2(x.data_ptr(), _ptr(_f32(bias)), B, C, H, W, bf, y.data_ptr(), _stream()),
               "ipsr_bias_relu_pool2")
    return y


CONV_FWD, CONV_BWD_DATA, CONVT_FWD, CONVT_BWD_DATA = 0, 1, 2, 3

# arithmetic of the Winograd GEMMs (include/ipsr_hip.h, ipsr_conv3x3_winograd_mp): "fp32" = fp32 operands on the fp32 MFMA (the
# reference's arithmetic, default); "bf16x3" / "bf16x6" = transformed operands split into 2 / 3 bf16 numbers, multiplied on the bf16
# MFMA with fp32 accumulation (error ~1e-4 / ~1e-5 of the output scale; a plain bf16 convolution: ~2e-3)
MATH_CODE = {None: 0, "fp32": 0, "bf16x3": 2, "bf16x6": 3}


def _act(t, name):
    """activation operand: contiguous fp32 or bf16 on the current device -> (tensor, is_bf16)"""
    if not torch.is_tensor(t) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor" % name)
    _on_current_device(t, name)
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("%s must be float32 or bfloat16, got %s" % (name, t.dtype))
    return (t if t.is_contiguous() else t.contiguous()), t.dtype == torch.bfloat16


def _io_code(in_bf16, out_dtype):
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("out_dtype must be float32 or bfloat16")
    return int(bool(in_bf16)) | (2 if out_dtype == torch.bfloat16 else 0)


def conv_out_dim(op, n, k, stride, pad, dil):
    if op in (CONV_FWD, CONV_BWD_DATA):
        return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1
    return (n - 1) * stride - 2 * pad + dil * (k - 1) + 1


def conv2d_supported(op, B, Cin, H, W, Cout, k, stride, pad, dil):
    """True when ipsr_conv2d implements this geometry (its workspace query answers 0 otherwise)."""
    return _lib.lib().ipsr_conv2d_workspace_bytes(op, B, Cin, H, W, Cout, k, stride, pad, dil) > 0


def conv2d(op, inp, weight, in_shape, Cout, k, stride, pad, dil):
    """The bias-free convolution kernels (include/ipsr_hip.h, ipsr_conv2d).  `in_shape` = (B, Cin, H, W) of the MODULE's input
    whatever the op; `inp` is x for the forward ops and dy for the backward-data ops.  Returns y / dx (fp32, contiguous)."""
    B, Cin, H, W = in_shape
    Ho, Wo = conv_out_dim(op, H, k, stride, pad, dil), conv_out_dim(op, W, k, stride, pad, dil)
    inp = _req(inp, torch.float32, "conv input")
    weight = _req(weight, torch.float32, "conv weight")
    fwd = op in (CONV_FWD, CONVT_FWD)
    want_in = (B, Cin, H, W) if fwd else (B, Cout, Ho, Wo)
    want_w = (Cout, Cin, k, k) if op in (CONV_FWD, CONV_BWD_DATA) else (Cin, Cout, k, k)
    if tuple(inp.shape) != want_in or tuple(weight.shape) != want_w:
        raise RuntimeError("conv2d op %d: input %s / weight %s do not match %s / %s" % (op, tuple(inp.shape), tuple(weight.shape), want_in, want_w))
    out = torch.empty((B, Cout, Ho, Wo) if fwd else (B, Cin, H, W), dtype=torch.float32, device=inp.device)
    L = _lib.lib()
    nbytes = L.ipsr_conv2d_workspace_bytes(op, B, Cin, H, W, Cout, k, stride, pad, dil)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv2d: op %d with k=%d stride=%d pad=%d dil=%d, Cin=%d Cout=%d is not implemented: %s"
                                  % (op, k, stride, pad, dil, Cin, Cout, L.ipsr_last_error().decode("utf-8", "replace")))
    ws = _workspace(nbytes, inp.device)
    _lib.check(L.ipsr_conv2d(op, inp.data_ptr(), weight.data_ptr(), out.data_ptr(), B, Cin, H, W, Cout, k, stride, pad, dil,
                             ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv2d")
    return out


def winograd_supported(op, B, Cin, H, W, Cout):
    return _lib.lib().ipsr_conv3x3_winograd_workspace_bytes(op, B, Cin, H, W, Cout) > 0


_EPILOGUE = {None: 0, "none": 0, "relu": 1, "relu_pool": 2}


def winograd_filter_cache(op, Cin, Cout, device):
    """An empty buffer for the transformed filter of one layer (see conv3x3_winograd's `filter_cache`)."""
    n = _lib.lib().ipsr_conv3x3_winograd_filter_floats(op, Cin, Cout)
    return torch.empty(n, dtype=torch.float32, device=device)


def conv3x3_winograd(op, inp, weight, in_shape, Cout, bias=None, epilogue=None, filter_cache=None, filter_cache_valid=False,
                     math=None, out_dtype=None):
    """k3 s1 p1 convolution / transposed convolution / their input gradients by Winograd F(4x4,3x3) (ipsr_conv3x3_winograd_mp).
    epilogue: None | "relu" (out = relu(conv + bias)) | "relu_pool" (... followed by the 2x2 max-pool; output [.,.,H/2,W/2]);
    filter_cache: buffer from `winograd_filter_cache`; the transformed filter is written into it unless filter_cache_valid (a
    cache is only valid for the `math` it was filled under);
    math: see MATH_CODE; inp may be fp32 or bf16, out_dtype (default: inp's dtype) likewise."""
    B, Cin, H, W = in_shape
    inp, in_bf = _act(inp, "conv input")
    weight = _req(weight, torch.float32, "conv weight")
    out_dtype = out_dtype or inp.dtype
    fwd = op in (CONV_FWD, CONVT_FWD)
    want_in = (B, Cin, H, W) if fwd else (B, Cout, H, W)
    want_w = (Cout, Cin, 3, 3) if op in (CONV_FWD, CONV_BWD_DATA) else (Cin, Cout, 3, 3)
    if tuple(inp.shape) != want_in or tuple(weight.shape) != want_w:
        raise RuntimeError("conv3x3_winograd op %d: input %s / weight %s do not match %s / %s" % (op, tuple(inp.shape), tuple(weight.shape), want_in, want_w))
    epi = _EPILOGUE[epilogue]
    if epi and not fwd:
        raise ValueError("conv3x3_winograd: an epilogue only makes sense on a forward op")
    if epi == 2 and (H % 2 or W % 2):
        raise RuntimeError("conv3x3_winograd: relu_pool needs even extents, got %dx%d" % (H, W))
    kout = Cout if fwd else Cin
    oshape = (B, kout, H // 2, W // 2) if epi == 2 else (B, kout, H, W)
    out = torch.empty(oshape, dtype=out_dtype, device=inp.device)
    L = _lib.lib()
    nbytes = L.ipsr_conv3x3_winograd_workspace_bytes(op, B, Cin, H, W, Cout)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv3x3_winograd: op %d Cin=%d Cout=%d is not implemented" % (op, Cin, Cout))
    if filter_cache is not None and filter_cache.numel() != L.ipsr_conv3x3_winograd_filter_floats(op, Cin, Cout):
        raise RuntimeError("conv3x3_winograd: filter_cache has the wrong size")
    ws = _workspace(nbytes, inp.device)
    _lib.check(L.ipsr_conv3x3_winograd_mp(op, inp.data_ptr(), weight.data_ptr(), _ptr(_f32(bias)), epi, _ptr(filter_cache),
                                          int(bool(filter_cache_valid)), out.data_ptr(), B, Cin, H, W, Cout, MATH_CODE[math], _io_code(in_bf, out_dtype),
                                          ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv3x3_winograd_mp")
    return out


def conv3x3_bf16_supported(op, B, Cin, H, W, Cout):
    return _lib.lib().ipsr_conv3x3_bf16_workspace_bytes(op, B, Cin, H, W, Cout) > 0


# frozen weights: weight tensor (weakly held) -> {op: (version, shape, buffer holding the re-packed bf16 weights)}.  Keyed by the tensor OBJECT:
# a storage pointer comes back to life with another tensor's data once the first is freed (two test cases with equal shapes met that way).
# (id-keyed with a weak reference for liveness: tensors compare elementwise, which rules out a WeakKeyDictionary)
_BF16_PACKS = {}


def _packs_of(weight):
    ent = _BF16_PACKS.get(id(weight))
    if ent is None or ent[0]() is not weight:
        key = id(weight)
        ent = (weakref.ref(weight, lambda _r, key=key: _BF16_PACKS.pop(key, None)), {})
        _BF16_PACKS[key] = ent
    return ent[1]


def conv3x3_bf16(op, inp, weight, in_shape, Cout, out_dtype=torch.bfloat16, keep_packed=False):
    """k3 s1 p1 convolution / transposed convolution / their input gradients as ONE direct implicit GEMM on the bf16 matrix cores
    (ipsr_conv3x3_bf16, csrc/conv_bf16.hip): bf16 activations in, bf16 or fp32 out, fp32 weights cast inside.  BASELINE config 5.
    keep_packed: the weights are FROZEN (VGG16): their re-packed bf16 image is kept (keyed by storage pointer and version) and the
    packing launch skipped from the second call on."""
    B, Cin, H, W = in_shape
    inp, in_bf = _act(inp, "conv input")
    if not in_bf:
        raise TypeError("conv3x3_bf16 reads bf16 activations, got %s" % inp.dtype)
    weight = _req(weight, torch.float32, "conv weight")
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("conv3x3_bf16 writes bf16 or fp32, not %s" % out_dtype)
    fwd = op in (CONV_FWD, CONVT_FWD)
    want_in = (B, Cin, H, W) if fwd else (B, Cout, H, W)
    want_w = (Cout, Cin, 3, 3) if op in (CONV_FWD, CONV_BWD_DATA) else (Cin, Cout, 3, 3)
    if tuple(inp.shape) != want_in or tuple(weight.shape) != want_w:
        raise RuntimeError("conv3x3_bf16 op %d: input %s / weight %s do not match %s / %s" % (op, tuple(inp.shape), tuple(weight.shape), want_in, want_w))
    L = _lib.lib()
    nbytes = L.ipsr_conv3x3_bf16_workspace_bytes(op, B, Cin, H, W, Cout)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv3x3_bf16: op %d on %s is not implemented (%s)" % (op, (B, Cin, H, W, Cout), _lib.lib().ipsr_last_error().decode("utf-8", "replace")))
    out = torch.empty((B, Cout if fwd else Cin, H, W), dtype=out_dtype, device=inp.device)
    valid = 0
    if keep_packed:
        packs = _packs_of(weight)
        ent = packs.get(op)
        if ent is not None and ent[0] == (weight._version, weight.data_ptr()) and ent[1] == tuple(weight.shape) and ent[2].numel() >= nbytes \
                and ent[2].device == inp.device:
            ws, valid = ent[2], 1
        else:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=inp.device)
            packs[op] = ((weight._version, weight.data_ptr()), tuple(weight.shape), ws)
    else:
        ws = _workspace(nbytes, inp.device)
    _lib.check(L.ipsr_conv3x3_bf16_packed(op, inp.data_ptr(), weight.data_ptr(), out.data_ptr(), B, Cin, H, W, Cout, int(out_dtype == torch.bfloat16),
                                          valid, ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv3x3_bf16")
    return out


def conv4x4s2_bf16_supported(mode, B, Kc, Cf, nh, nw):
    return mode in (S2_FINE_TO_COARSE, S2_COARSE_TO_FINE) and _lib.lib().ipsr_conv4x4s2_bf16_workspace_bytes(mode, B, Kc, Cf, nh, nw) > 0


def conv4x4s2_bf16(mode, inp, weight, B, Kc, Cf, nh, nw, out_dtype=torch.bfloat16):
    """The k4 s2 p1 layers as ONE direct implicit GEMM on the bf16 matrix cores (ipsr_conv4x4s2_bf16), in the coarse / fine terms of
    conv4x4s2_winograd: mode S2_FINE_TO_COARSE: inp = fine [B,Cf,2nh,2nw] -> coarse [B,Kc,nh,nw]; S2_COARSE_TO_FINE: the reverse.
    weight [Kc,Cf,4,4] fp32 (Conv2d: [Cout,Cin]; ConvTranspose2d: [Cin,Cout])."""
    inp, in_bf = _act(inp, "conv input")
    if not in_bf:
        raise TypeError("conv4x4s2_bf16 reads bf16 activations, got %s" % inp.dtype)
    weight = _req(weight, torch.float32, "conv weight")
    if mode not in (S2_FINE_TO_COARSE, S2_COARSE_TO_FINE):
        raise ValueError("conv4x4s2_bf16: mode %r" % (mode,))
    want_in = (B, Cf, 2 * nh, 2 * nw) if mode == S2_FINE_TO_COARSE else (B, Kc, nh, nw)
    if tuple(inp.shape) != want_in or tuple(weight.shape) != (Kc, Cf, 4, 4):
        raise RuntimeError("conv4x4s2_bf16 mode %d: input %s / weight %s do not match %s / %s" % (mode, tuple(inp.shape), tuple(weight.shape), want_in, (Kc, Cf, 4, 4)))
    L = _lib.lib()
    nbytes = L.ipsr_conv4x4s2_bf16_workspace_bytes(mode, B, Kc, Cf, nh, nw)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv4x4s2_bf16: mode %d on %s is not implemented (%s)" % (mode, (B, Kc, Cf, nh, nw), L.ipsr_last_error().decode("utf-8", "replace")))
    oshape = (B, Kc, nh, nw) if mode == S2_FINE_TO_COARSE else (B, Cf, 2 * nh, 2 * nw)
    out = torch.empty(oshape, dtype=out_dtype, device=inp.device)
    ws = _workspace(nbytes, inp.device)
    _lib.check(L.ipsr_conv4x4s2_bf16(mode, inp.data_ptr(), weight.data_ptr(), out.data_ptr(), B, Kc, Cf, nh, nw, int(out_dtype == torch.bfloat16),
                                     ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv4x4s2_bf16")
    return out


def conv4x4s2_bf16_wrw_supported(B, Kc, Cf, nh, nw):
    return _lib.lib().ipsr_conv4x4s2_bf16_wrw_workspace_bytes(B, Kc, Cf, nh, nw) > 0


def conv4x4s2_bf16_wrw(fine, coarse, B, Kc, Cf, nh, nw, out=None):
    """Weight gradient [Kc,Cf,4,4] (fp32) of a k4 s2 p1 layer from its bf16 fine [B,Cf,2nh,2nw] and coarse [B,Kc,nh,nw] tensors
    (ipsr_conv4x4s2_bf16_wrw); `out`: optional contiguous fp32 destination (a gradient bucket slice)."""
    fine, f_bf = _act(fine, "fine tensor")
    coarse, c_bf = _act(coarse, "coarse tensor")
    if not (f_bf and c_bf):
        raise TypeError("conv4x4s2_bf16_wrw reads bf16 tensors")
    if tuple(fine.shape) != (B, Cf, 2 * nh, 2 * nw) or tuple(coarse.shape) != (B, Kc, nh, nw):
        raise RuntimeError("conv4x4s2_bf16_wrw: fine %s / coarse %s do not match %s / %s" % (tuple(fine.shape), tuple(coarse.shape), (B, Cf, 2 * nh, 2 * nw), (B, Kc, nh, nw)))
    shape = (Kc, Cf, 4, 4)
    if out is not None and (tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != fine.device):
        raise RuntimeError("conv4x4s2_bf16_wrw: `out` must be a contiguous fp32 %s tensor on %s" % (shape, fine.device))
    L = _lib.lib()
    nbytes = L.ipsr_conv4x4s2_bf16_wrw_workspace_bytes(B, Kc, Cf, nh, nw)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv4x4s2_bf16_wrw: %s is not implemented (%s)" % ((B, Kc, Cf, nh, nw), L.ipsr_last_error().decode("utf-8", "replace")))
    dw = out if out is not None else torch.empty(shape, dtype=torch.float32, device=fine.device)
    ws = _workspace(nbytes, fine.device)
    _lib.check(L.ipsr_conv4x4s2_bf16_wrw(fine.data_ptr(), coarse.data_ptr(), dw.data_ptr(), B, Kc, Cf, nh, nw, ws.data_ptr(), ws.numel(), _stream()),
               "ipsr_conv4x4s2_bf16_wrw")
    return dw


def conv3x3_bf16_wrw_supported(transposed, B, Cin, H, W, Cout):
    return _lib.lib().ipsr_conv3x3_bf16_wrw_workspace_bytes(int(transposed), B, Cin, H, W, Cout) > 0


def conv3x3_bf16_wrw(transposed, x, dy, Cout, out=None):
    """Weight gradient of a k3 s1 p1 Conv2d (-> [Cout,Cin,3,3]) / ConvTranspose2d (-> [Cin,Cout,3,3]) on the bf16 matrix cores
    (ipsr_conv3x3_bf16_wrw): x, dy bf16; the result fp32 (optionally written into `out`, e.g. a slice of a gradient bucket)."""
    x, x_bf = _act(x, "conv input")
    dy, dy_bf = _act(dy, "grad_output")
    if not (x_bf and dy_bf):
        raise TypeError("conv3x3_bf16_wrw reads bf16 tensors")
    B, Cin, H, W = x.shape
    if tuple(dy.shape) != (B, Cout, H, W):
        raise RuntimeError("conv3x3_bf16_wrw: grad_output %s does not match %s" % (tuple(dy.shape), (B, Cout, H, W)))
    shape = (Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3)
    if out is not None and (tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device):
        raise RuntimeError("conv3x3_bf16_wrw: `out` must be a contiguous fp32 %s tensor on %s" % (shape, x.device))
    L = _lib.lib()
    nbytes = L.ipsr_conv3x3_bf16_wrw_workspace_bytes(int(transposed), B, Cin, H, W, Cout)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv3x3_bf16_wrw: %s is not implemented (%s)" % ((B, Cin, H, W, Cout), L.ipsr_last_error().decode("utf-8", "replace")))
    dw = out if out is not None else torch.empty(shape, dtype=torch.float32, device=x.device)
    ws = _workspace(nbytes, x.device)
    _lib.check(L.ipsr_conv3x3_bf16_wrw(int(transposed), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout, ws.data_ptr(), ws.numel(),
                                       _stream()), "ipsr_conv3x3_bf16_wrw")
    return dw


GEOM_K4_S2_P3_D2 = 0       # Conv2d(k4, stride 2, pad 3, dilation 2): netG's down convolution
GEOM_K4_S1_P1 = 1          # Conv2d(k4, stride 1, pad 1): netD's fourth convolution


def conv4x4_geometry(k, stride, pad, dil):
    """-> the `geom` of ipsr_conv4x4_winograd for a Conv2d, or None."""
    if (k, stride, pad, dil) == (4, 2, 3, 2):
        return GEOM_K4_S2_P3_D2
    if (k, stride, pad, dil) == (4, 1, 1, 1):
        return GEOM_K4_S1_P1
    return None


def dilated_winograd_supported(mode, B, Cin, H, W, Cout, geom=GEOM_K4_S2_P3_D2):
    return _lib.lib().ipsr_conv4x4_winograd_workspace_bytes(geom, mode, B, Cin, H, W, Cout) > 0


def conv4x4_dilated_winograd(mode, a, b, in_shape, Cout, out=None, geom=GEOM_K4_S2_P3_D2, math=None, out_dtype=None):
    """Conv2d(k4, stride 2, pad 3, dilation 2) (geom 0) or Conv2d(k4, stride 1, pad 1) (geom 1) by Winograd F(3x3,4x4)
    (ipsr_conv4x4_winograd_mp): mode 0 forward (a = x, b = weight -> y), 1 backward-data (a = dy, b = weight -> dx), 2 weight
    gradient (a = x, b = dy -> dw, always fp32).  Activations fp32 or bf16; math: see MATH_CODE."""
    B, Cin, H, W = in_shape
    a, a_bf = _act(a, "operand a")
    if mode == 2:
        b, b_bf = _act(b, "operand b")
        if b_bf != a_bf:
            raise TypeError("conv4x4_winograd: x and dy of a weight gradient must have the same dtype")
    else:
        b = _req(b, torch.float32, "weight")
    res_dtype = torch.float32 if mode == 2 else (out_dtype or a.dtype)
    Ho, Wo = (H // 2, W // 2) if geom == GEOM_K4_S2_P3_D2 else (H - 1, W - 1)
    xs, ys, wsh = (B, Cin, H, W), (B, Cout, Ho, Wo), (Cout, Cin, 4, 4)
    want = {0: (xs, wsh, ys), 1: (ys, wsh, xs), 2: (xs, ys, wsh)}[mode]
    if tuple(a.shape) != want[0] or tuple(b.shape) != want[1]:
        raise RuntimeError("conv4x4_winograd mode %d: operands %s / %s do not match %s / %s" % (mode, tuple(a.shape), tuple(b.shape), want[0], want[1]))
    if out is not None and (tuple(out.shape) != tuple(want[2]) or out.dtype != res_dtype or not out.is_contiguous() or out.device != a.device):
        raise RuntimeError("conv4x4_winograd: `out` must be a contiguous %s %s tensor on %s" % (res_dtype, tuple(want[2]), a.device))
    if out is None:
        out = torch.empty(want[2], dtype=res_dtype, device=a.device)
    L = _lib.lib()
    nbytes = L.ipsr_conv4x4_winograd_workspace_bytes(geom, mode, B, Cin, H, W, Cout)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv4x4_winograd: geometry %d mode %d Cin=%d Cout=%d %dx%d is not implemented" % (geom, mode, Cin, Cout, H, W))
    ws = _workspace(nbytes, a.device)
    _lib.check(L.ipsr_conv4x4_winograd_mp(geom, mode, a.data_ptr(), b.data_ptr(), out.data_ptr(), B, Cin, H, W, Cout, MATH_CODE[math],
                                          _io_code(a_bf, res_dtype), ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv4x4_winograd_mp")
    return out


S2_FINE_TO_COARSE, S2_COARSE_TO_FINE, S2_WEIGHT_GRAD = 0, 1, 2


def s2_winograd_supported(mode, B, Kc, Cf, nh, nw):
    return _lib.lib().ipsr_conv4x4s2_winograd_workspace_bytes(mode, B, Kc, Cf, nh, nw) > 0


def conv4x4s2_winograd(mode, a, b, B, Kc, Cf, nh, nw, out=None, math=None, out_dtype=None):
    """The k4 stride-2 pad-1 layers by Winograd F(5x5,2x2) on the polyphase components (ipsr_conv4x4s2_winograd_mp).
    fine = [B,Cf,2nh,2nw] (x of Conv2d, y of ConvTranspose2d), coarse = [B,Kc,nh,nw], weight = [Kc,Cf,4,4].
    mode 0 fine -> coarse (a = fine, b = weight), 1 coarse -> fine (a = coarse, b = weight), 2 weight gradient (a = fine, b = coarse;
    result fp32).  Activations fp32 or bf16; math: see MATH_CODE."""
    a, a_bf = _act(a, "operand a")
    if mode == 2:
        b, b_bf = _act(b, "operand b")
        if b_bf != a_bf:
            raise TypeError("conv4x4s2_winograd: both operands of a weight gradient must have the same dtype")
    else:
        b = _req(b, torch.float32, "weight")
    res_dtype = torch.float32 if mode == 2 else (out_dtype or a.dtype)
    fine, coarse, wsh = (B, Cf, 2 * nh, 2 * nw), (B, Kc, nh, nw), (Kc, Cf, 4, 4)
    want = {0: (fine, wsh, coarse), 1: (coarse, wsh, fine), 2: (fine, coarse, wsh)}[mode]
    if tuple(a.shape) != want[0] or tuple(b.shape) != want[1]:
        raise RuntimeError("conv4x4s2_winograd mode %d: operands %s / %s do not match %s / %s" % (mode, tuple(a.shape), tuple(b.shape), want[0], want[1]))
    if out is not None and (tuple(out.shape) != tuple(want[2]) or out.dtype != res_dtype or not out.is_contiguous() or out.device != a.device):
        raise RuntimeError("conv4x4s2_winograd: `out` must be a contiguous %s %s tensor on %s" % (res_dtype, tuple(want[2]), a.device))
    if out is None:
        out = torch.empty(want[2], dtype=res_dtype, device=a.device)
    L = _lib.lib()
    nbytes = L.ipsr_conv4x4s2_winograd_workspace_bytes(mode, B, Kc, Cf, nh, nw)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv4x4s2_winograd: mode %d Kc=%d Cf=%d %dx%d is not implemented" % (mode, Kc, Cf, nh, nw))
    ws = _workspace(nbytes, a.device)
    _lib.check(L.ipsr_conv4x4s2_winograd_mp(mode, a.data_ptr(), b.data_ptr(), out.data_ptr(), B, Kc, Cf, nh, nw, MATH_CODE[math],
                                            _io_code(a_bf, res_dtype), ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv4x4s2_winograd_mp")
    return out


def thin_supported(op, Cin, H, W, Cout):
    """3x3 stride-1 pad-1 with 3 or 6 channels on one side (ipsr_conv3x3_thin): which side is thin depends on the operation."""
    fwd = op in (CONV_FWD, CONVT_FWD)
    i, o = (Cin, Cout) if fwd else (Cout, Cin)
    if i in (3, 6) and o % 16 == 0 and o >= 16:
        return W % 2 == 0
    return o in (3, 6) and W % 4 == 0 and i * o * 36 <= 48 * 1024 and i >= 16


def conv_to_one_supported(B, Cin, H, W, k, stride, pad, dil):
    """nn.Conv2d(Cin, 1, k, 1, pad) as ipsr_conv_to_one takes it (csrc/thin_conv.hip): a workspace probe, no kernel is launched."""
    if stride != 1 or dil != 1 or pad < 0 or k not in (3, 4):
        return False
    return _lib.lib().ipsr_conv_to_one_workspace_bytes(B, Cin, H, W, k, pad) > 0 and (2 * (H + 2 * pad) * (W + 2 * pad) + 2 * H * W + 64) * 4 <= 64 * 1024


def conv_to_one(x, w, pad):
    """y = conv2d(x, w, stride 1, padding pad) for w [1,C,K,K]: one pass over x (netD's last layer)."""
    x = _req(x, torch.float32, "input")
    w = _req(w, torch.float32, "weight")
    B, C, H, W = x.shape
    K = int(w.shape[-1])
    if tuple(w.shape) != (1, C, K, K):
        raise RuntimeError("conv_to_one: weight %s does not match %d input channels / one output" % (tuple(w.shape), C))
    y = torch.empty((B, 1, H + 2 * pad - K + 1, W + 2 * pad - K + 1), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _workspace(L.ipsr_conv_to_one_workspace_bytes(B, C, H, W, K, int(pad)), x.device)
    _lib.check(L.ipsr_conv_to_one(0, x.data_ptr(), w.data_ptr(), y.data_ptr(), B, C, H, W, K, int(pad), ws.data_ptr(), ws.numel(), _stream()),
               "ipsr_conv_to_one")
    return y


def conv_to_one_wrw(x, dy, K, pad, out=None):
    """dW [1,C,K,K] of the same layer: one pass over x."""
    x = _req(x, torch.float32, "input")
    dy = _req(dy, torch.float32, "grad_output")
    B, C, H, W = x.shape
    if tuple(dy.shape) != (B, 1, H + 2 * pad - K + 1, W + 2 * pad - K + 1):
        raise RuntimeError("conv_to_one_wrw: grad_output %s does not match input %s, k=%d, pad=%d" % (tuple(dy.shape), tuple(x.shape), K, pad))
    dw = out if out is not None else torch.empty((1, C, K, K), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ipsr_conv_to_one(2, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, C, H, W, K, int(pad), None, 0, _stream()),
               "ipsr_conv_to_one")
    return dw


def conv3x3_thin(op, inp, weight, in_shape, Cout, bias=None, relu=False, out=None, out_dtype=None):
    """k3 s1 p1 Conv2d / ConvTranspose2d forward or backward-data with a 3- or 6-channel side (ipsr_conv3x3_thin_io).  `in_shape` =
    the module's input (B, Cin, H, W); `inp` is x for the forward ops and dy for the backward-data ops; bias / ReLU only few -> many.
    fp32 or bf16 activation tensors on either side (`out_dtype`, default = the input's); with a bf16 side the arithmetic is autocast's
    (bf16 operands, fp32 accumulation)."""
    B, Cin, H, W = in_shape
    inp, _ = _act(inp, "input")
    weight = _req(weight, torch.float32, "weight")
    out_dtype = out_dtype or (out.dtype if out is not None else inp.dtype)
    fwd = op in (CONV_FWD, CONVT_FWD)
    I, O = (Cin, Cout) if fwd else (Cout, Cin)
    if tuple(inp.shape) != (B, I, H, W):
        raise RuntimeError("conv3x3_thin: input %s does not match %s" % (tuple(inp.shape), (B, I, H, W)))
    want_w = (Cout, Cin, 3, 3) if op in (CONV_FWD, CONV_BWD_DATA) else (Cin, Cout, 3, 3)
    if tuple(weight.shape) != want_w:
        raise RuntimeError("conv3x3_thin op %d: weight %s does not match %s" % (op, tuple(weight.shape), want_w))
    if not thin_supported(op, Cin, H, W, Cout):
        raise NotImplementedError("conv3x3_thin op %d: Cin=%d Cout=%d %dx%d is not a thin-layer shape" % (op, Cin, Cout, H, W))
    so, si, flip = {CONV_FWD: (Cin * 9, 9, 0), CONV_BWD_DATA: (9, Cin * 9, 1), CONVT_FWD: (9, Cout * 9, 1), CONVT_BWD_DATA: (Cout * 9, 9, 0)}[op]
    io = _io_code(inp.dtype == torch.bfloat16, out_dtype)
    if out is None:
        out = torch.empty((B, O, H, W), dtype=out_dtype, device=inp.device)
    elif tuple(out.shape) != (B, O, H, W) or out.dtype != out_dtype or not out.is_contiguous() or out.device != inp.device:
        raise RuntimeError("conv3x3_thin: `out` must be a contiguous %s %s tensor on %s" % (out_dtype, (B, O, H, W), inp.device))
    few2many = I in (3, 6) and O % 16 == 0
    _lib.check(_lib.lib().ipsr_conv3x3_thin_io(0 if few2many else 1, inp.data_ptr(), weight.data_ptr(), _ptr(_f32(bias)) if bias is not None else None,
                                               int(bool(relu)), out.data_ptr(), B, I, O, H, W, so, si, flip, io, _stream()), "ipsr_conv3x3_thin_io")
    return out


def conv3x3_thin_wrw(transposed, x, dy, out=None):
    """Weight gradient (fp32) of a k3 s1 p1 Conv2d ([Cout,Cin,3,3]) / ConvTranspose2d ([Cin,Cout,3,3]) with a 3- or 6-channel side; x and dy
    each fp32 or bf16."""
    x, _ = _act(x, "x")
    dy, _ = _act(dy, "dy")
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    wshape = (Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3)
    big, small = (x, dy) if transposed else (dy, x)            # G[big ch][small ch] is the weight's own [dim0][dim1] only when dim0 is wide
    Cb, Cs = big.shape[1], small.shape[1]
    L = _lib.lib()
    if Cs in (3, 6):
        g = out if out is not None else torch.empty(wshape, dtype=torch.float32, device=x.device)
        nbytes = L.ipsr_conv3x3_thin_wrw_workspace_bytes(B, Cb, Cs, H, W)
        if nbytes == 0:
            raise NotImplementedError("ipsr_conv3x3_thin_wrw: Cb=%d Cs=%d %dx%d is not implemented" % (Cb, Cs, H, W))
        ws = _workspace(nbytes, x.device)
        io = int(big.dtype == torch.bfloat16) | (2 if small.dtype == torch.bfloat16 else 0)
        _lib.check(L.ipsr_conv3x3_thin_wrw_io(big.data_ptr(), small.data_ptr(), g.data_ptr(), B, Cb, Cs, H, W, io, ws.data_ptr(), ws.numel(), _stream()),
                   "ipsr_conv3x3_thin_wrw_io")
        return g
    raise NotImplementedError("conv3x3_thin_wrw: the weight's first channel dimension must be the wide one (got %s)" % (wshape,))


def _thin_f2m_geometry(op, B, Cin, H, W, Cout, k, stride):
    """(Cs, O, Ho, Wo, so, si, flip) of a few -> many pass in the terms of ipsr_conv_thin_f2m_mfma, or None: Conv2d forward (narrow input
    -> wide output on the strided grid) and ConvTranspose2d input gradient (narrow dy on the fine grid -> wide dx on the coarse one)."""
    T = k * k
    if op == CONV_FWD:
        if H % stride or W % stride:
            return None
        return Cin, Cout, H // stride, W // stride, Cin * T, T, 0
    if op == CONVT_BWD_DATA:                                   # dx[ci][iy][ix] = sum_{co,t} W[ci][co][t] dy[co][iy*st + r - pad][ix*st + s - pad]
        return Cout, Cin, H, W, Cout * T, T, 0
    return None


def thin_f2m_mfma_supported(op, B, Cin, H, W, Cout, k, stride):
    g = _thin_f2m_geometry(op, B, Cin, H, W, Cout, k, stride)
    return g is not None and bool(_lib.lib().ipsr_conv_thin_f2m_mfma_supported(B, g[0], g[1], g[2], g[3], k, stride))


def conv_thin_f2m_mfma(op, inp, weight, in_shape, Cout, k, stride, bias=None, relu=False, out_dtype=torch.bfloat16):
    """Conv2d forward / ConvTranspose2d input gradient with 3 or 6 channels on the narrow (read) side on the bf16 matrix cores
    (ipsr_conv_thin_f2m_mfma): k3 s1 p1 or k4 s2 p1.  `in_shape` = the module's input (B, Cin, H, W); `inp` = x (forward) or dy (input
    gradient), fp32 or bf16 (rounded to bf16 inside); bias / ReLU in fp32; bf16 or fp32 out."""
    B, Cin, H, W = in_shape
    inp, in_bf = _act(inp, "input")
    weight = _req(weight, torch.float32, "weight")
    g = _thin_f2m_geometry(op, B, Cin, H, W, Cout, k, stride)
    if g is None or not _lib.lib().ipsr_conv_thin_f2m_mfma_supported(B, g[0], g[1], g[2], g[3], k, stride):
        raise NotImplementedError("conv_thin_f2m_mfma op %d: Cin=%d Cout=%d %dx%d k%d s%d is not implemented" % (op, Cin, Cout, H, W, k, stride))
    Cs, O, Ho, Wo, so, si, flip = g
    want_in = (B, Cs, Ho * stride, Wo * stride)
    want_w = (Cout, Cin, k, k) if op == CONV_FWD else (Cin, Cout, k, k)
    if tuple(inp.shape) != want_in or tuple(weight.shape) != want_w:
        raise RuntimeError("conv_thin_f2m_mfma op %d: input %s / weight %s do not match %s / %s" % (op, tuple(inp.shape), tuple(weight.shape), want_in, want_w))
    out = torch.empty((B, O, Ho, Wo), dtype=out_dtype, device=inp.device)
    _lib.check(_lib.lib().ipsr_conv_thin_f2m_mfma(inp.data_ptr(), weight.data_ptr(), _ptr(_f32(bias)) if bias is not None else None, int(bool(relu)),
                                                  out.data_ptr(), B, Cs, O, Ho, Wo, k, stride, so, si, flip, _io_code(in_bf, out_dtype), _stream()),
               "ipsr_conv_thin_f2m_mfma")
    return out


def thin_wrw_mfma_supported(transposed, B, Cin, H, W, Cout, k, stride):
    """Weight gradient of a k3 s1 p1 / k4 s2 p1 layer with 3 or 6 channels on its NARROW side on the bf16 matrix cores: the wide side
    must be the weight's first dimension (Conv2d: Cout wide; ConvTranspose2d: Cin wide) and live on the coarse grid."""
    if transposed:
        Kb, Cs, Hb, Wb = Cin, Cout, H, W                   # big = x on the module's input grid (the coarse one)
    else:
        if H % stride or W % stride:
            return False
        Kb, Cs, Hb, Wb = Cout, Cin, H // stride, W // stride
    return _lib.lib().ipsr_conv_thin_wrw_mfma_workspace_bytes(B, Kb, Cs, Hb, Wb, k, stride) > 0


def conv_thin_wrw_mfma(transposed, x, dy, k, stride, out=None):
    """-> dW (fp32, the module's layout) of a thin Conv2d / ConvTranspose2d on the matrix cores (ipsr_conv_thin_wrw_mfma).  A bf16 wide
    tensor multiplies in bf16 (the narrow one may be fp32: it is rounded inside, as autocast's cast would); fp32 tensors multiply in fp32."""
    x, _ = _act(x, "x")
    dy, _ = _act(dy, "dy")
    big, small = (x, dy) if transposed else (dy, x)
    if big.dtype == torch.float32 and small.dtype != torch.float32:
        raise TypeError("conv_thin_wrw_mfma: an fp32 wide tensor needs an fp32 narrow one")
    B, Kb, Hb, Wb = big.shape
    Cs = small.shape[1]
    if tuple(small.shape) != (B, Cs, Hb * stride, Wb * stride):
        raise RuntimeError("conv_thin_wrw_mfma: narrow tensor %s does not match wide %s at stride %d" % (tuple(small.shape), tuple(big.shape), stride))
    L = _lib.lib()
    nbytes = L.ipsr_conv_thin_wrw_mfma_workspace_bytes(B, Kb, Cs, Hb, Wb, k, stride)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv_thin_wrw_mfma: Kb=%d Cs=%d %dx%d k%d s%d is not implemented" % (Kb, Cs, Hb, Wb, k, stride))
    g = out if out is not None else torch.empty((Kb, Cs, k, k), dtype=torch.float32, device=x.device)
    if tuple(g.shape) != (Kb, Cs, k, k) or g.dtype != torch.float32 or not g.is_contiguous():
        raise RuntimeError("conv_thin_wrw_mfma: `out` must be a contiguous fp32 %s tensor" % ((Kb, Cs, k, k),))
    ws = _workspace(nbytes, x.device)
    _lib.check(L.ipsr_conv_thin_wrw_mfma(big.data_ptr(), small.data_ptr(), g.data_ptr(), B, Kb, Cs, Hb, Wb, k, stride, int(big.dtype == torch.bfloat16) | (2 if small.dtype == torch.bfloat16 else 0),
                                         ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv_thin_wrw_mfma")
    return g


SM_DATA, SM_WRW, SM_FWD = 0, 1, 2


def smallmap_supported(op, B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil):
    return _lib.lib().ipsr_conv_smallmap_workspace_bytes(op, B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil) > 0


def conv_smallmap(op, a, b, B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil, out=None):
    """Small-map convolutions with the weight tensor [R,Cq,k,k] as the GEMM operand in place (ipsr_conv_smallmap):
    op SM_DATA: a = in [B,R,Ho,Wo], b = weight -> [B,Cq,Hf,Wf] (Conv2d backward-data / ConvTranspose2d forward);
    op SM_WRW:  a = coarse [B,R,Ho,Wo], b = fine [B,Cq,Hf,Wf] -> dW [R,Cq,k,k];
    op SM_FWD:  a = fine [B,Cq,Hf,Wf], b = weight -> [B,R,Ho,Wo] (Conv2d forward / ConvTranspose2d backward-data)."""
    a = _req(a, torch.float32, "operand a")
    b = _req(b, torch.float32, "operand b")
    coarse, fine, wsh = (B, R, Ho, Wo), (B, Cq, Hf, Wf), (R, Cq, k, k)
    want = {SM_DATA: (coarse, wsh, fine), SM_WRW: (coarse, fine, wsh), SM_FWD: (fine, wsh, coarse)}[op]
    if tuple(a.shape) != want[0] or tuple(b.shape) != want[1]:
        raise RuntimeError("conv_smallmap op %d: operands %s / %s do not match %s / %s" % (op, tuple(a.shape), tuple(b.shape), want[0], want[1]))
    if out is not None and (tuple(out.shape) != tuple(want[2]) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != a.device):
        raise RuntimeError("conv_smallmap: `out` must be a contiguous fp32 %s tensor on %s" % (tuple(want[2]), a.device))
    if out is None:
        out = torch.empty(want[2], dtype=torch.float32, device=a.device)
    L = _lib.lib()
    nbytes = L.ipsr_conv_smallmap_workspace_bytes(op, B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil)
    if nbytes == 0:
        raise NotImplementedError("ipsr_conv_smallmap: op %d R=%d Cq=%d %dx%d -> %dx%d k%d s%d p%d d%d is not implemented" % (op, R, Cq, Hf, Wf, Ho, Wo, k, stride, pad, dil))
    ws = _workspace(nbytes, a.device)
    _lib.check(L.ipsr_conv_smallmap(op, a.data_ptr(), b.data_ptr(), out.data_ptr(), B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil,
                                    ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv_smallmap")
    return out


def conv3x3_winograd_wrw(transposed, x, dy, Cout, out=None, math=None):
    """Weight gradient of a k3 s1 p1 Conv2d (transposed=False -> [Cout,Cin,3,3]) / ConvTranspose2d (True -> [Cin,Cout,3,3]).
    out: optional contiguous fp32 tensor of that shape to write into (e.g. a slice of a gradient bucket).  x / dy fp32 or bf16 (both
    the same); the result is fp32; math: see MATH_CODE."""
    x, x_bf = _act(x, "conv input")
    dy, dy_bf = _act(dy, "grad_output")
    if x_bf != dy_bf:
        raise TypeError("conv3x3_winograd_wrw: x and dy must have the same dtype")
    B, Cin, H, W = x.shape
    if tuple(dy.shape) != (B, Cout, H, W):
        raise RuntimeError("conv3x3_winograd_wrw: grad_output %s does not match %s" % (tuple(dy.shape), (B, Cout, H, W)))
    shape = (Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3)
    if out is not None and (tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device):
        raise RuntimeError("conv3x3_winograd_wrw: `out` must be a contiguous fp32 %s tensor on %s" % (shape, x.device))
    dw = out if out is not None else torch.empty(shape, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _workspace(L.ipsr_conv3x3_winograd_wrw_workspace_bytes(int(transposed), B, Cin, H, W, Cout), x.device)
    _lib.check(L.ipsr_conv3x3_winograd_wrw_mp(int(transposed), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout, MATH_CODE[math],
                                              int(x_bf), ws.data_ptr(), ws.numel(), _stream()), "ipsr_conv3x3_winograd_wrw_mp")
    return dw


_IC_TICKETS = {}


def innercos_loss(x, cuse, mask_f32, target, strength, one_launch=False):
    """K9.  x [B,Cx,h,w] (only the first `cuse` channels are read), target [B,cuse,h,w] -> loss [] fp32."""
    x = _req(x, torch.float32, "in_data")
    target = _req(target, torch.float32, "target")
    mask = _req(mask_f32, torch.float32, "mask")
    B, Cx = x.shape[0], x.shape[1]
    N = x.shape[2] * x.shape[3]
    if tuple(target.shape) != (B, cuse, x.shape[2], x.shape[3]) or mask.numel() != N:
        raise RuntimeError("InnerCos: target %s / mask %s do not match input %s" % (tuple(target.shape), tuple(mask.shape), tuple(x.shape)))
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _workspace(L.innercos_workspace_bytes(B, cuse, N), x.device)
    if one_launch:
        # the last workgroup folds the block partials.  Its arrival counter is a word of OUR memory, one per (device, stream) — the calls
        # of a stream are ordered, and every launch leaves the word zero — allocated (zeroed) once.  Measured SLOWER than two launches
        # (csrc/innercos.hip): kept for the C-ABI's completeness and its test, not used by the modules.
        key = (x.device.index, _stream())
        ticket = _IC_TICKETS.get(key)
        if ticket is None:
            ticket = _IC_TICKETS[key] = torch.zeros(64, dtype=torch.int32, device=x.device)
        _lib.check(L.innercos_loss_fused(x.data_ptr(), B, Cx, cuse, N, mask.data_ptr(), target.data_ptr(), float(strength),
                                         loss.data_ptr(), ws.data_ptr(), ws.numel(), ticket.data_ptr(), _stream()), "innercos_loss_fused")
        return loss
    _lib.check(L.innercos_loss(x.data_ptr(), B, Cx, cuse, N, mask.data_ptr(), target.data_ptr(), float(strength),
                               loss.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "innercos_loss")
    return loss


def innercos_loss_backward(x, cuse, mask_f32, target, strength, grad_loss):
    x = _req(x, torch.float32, "in_data")
    target = _req(target, torch.float32, "target")
    mask = _req(mask_f32, torch.float32, "mask")
    gl = _req(grad_loss.reshape(1).to(torch.float32), torch.float32, "grad_loss")
    B, Cx = x.shape[0], x.shape[1]
    N = x.shape[2] * x.shape[3]
    gx = torch.empty_like(x)
    _lib.check(_lib.lib().innercos_loss_backward(x.data_ptr(), B, Cx, cuse, N, mask.data_ptr(), target.data_ptr(),
                                                 float(strength), gl.data_ptr(), gx.data_ptr(), _stream()),
               "innercos_loss_backward")
    return gx
