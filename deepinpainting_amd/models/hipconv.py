"""Convolutions of the four nets and of VGG16 on the hand-written HIP kernels, behind nn.Conv2d /
nn.ConvTranspose2d modules whose parameters, names and state_dict keys are untouched (models/networks.py:220-259,
404-432, 470-495, 510-515; models/vgg16.py:9-21).

Three engines per (operation, geometry), chosen by `select()` from measurements on MI355X at the step's shapes
(tools/bench_hipconv.py -> profiles/r02_hipconv_*.txt):

  "winograd"  csrc/winograd.hip   k3 s1 p1 forward / backward-data of Conv2d and ConvTranspose2d, F(4x4,3x3) on fp32 MFMA:
                                  2.0-2.4x MIOpen's F(2x2,3x3) assembly from 16x16 maps and 128 channels up
  "wino_dil"  csrc/winograd.hip   netG's dilated down convolution Conv2d(k4 s2 p3 d2) and netD's Conv2d(k4 s1 p1) by F(3x3,4x4) —
                                  forward, input and weight gradient, 1.4-2.0x MIOpen at 128..512 channels on 32x32..128x128 inputs
  "wino_s2"   csrc/winograd.hip   the 4x4 stride-2 pad-1 layers (Conv2d of netP/netD/netF, ConvTranspose2d of netP/netG) by F(5x5,2x2) on
                                  the polyphase components — forward, input and weight gradient, 5-30 % faster than MIOpen at
                                  >= 128 / 64 channels on coarse grids of 16..64
  "thin"      csrc/thin_conv.hip  3x3 stride-1 layers with a 3- or 6-channel side at full resolution (VGG conv1_1, netG's last ConvTranspose2d):
                                  one pass over the wide tensor on the vector ALUs, 1.3-3x MIOpen
  "thin_f2m"  csrc/thin_conv.hip  Conv2d 3 -> K, k4 s2 p1, forward under bf16 activations: the window gather on the bf16 matrix cores, one launch
  "thin_mfma" csrc/thin_conv.hip  weight gradient of the layers with 3 or 6 channels on the narrow side (k3 s1 p1, k4 s2 p1) under bf16
                                  activations: the pixel reduction on the bf16 matrix cores straight from NCHW, two launches
  "smallmap"  csrc/winograd.hip   the innermost levels: the weight tensor streamed once, 16 bytes per lane straight into MFMA operands —
                                  weight gradients of the 4x4 stride-2 layers up to 256 positions per batch (dW written in its native
                                  layout), forward and input gradient of the 3x3 / 4x4 layers up to 32 positions
  "direct"    csrc/conv_gemm.hip  one-launch implicit GEMM, NCHW in/out (every k3/k4, stride 1/2, dilated and transposed
                                  geometry of the nets, forward and backward-data): at parity with MIOpen (~100 TF), used
                                  where it measured >= 7 % faster
  "one"       csrc/thin_conv.hip  Conv2d with ONE output channel, stride 1 (netD's last layer, 512 -> 1 on 31x31): forward and weight
                                  gradient as one pass over the input
  "miopen"    torch               everything else
Weight gradients: Winograd F(3x3,4x4) (csrc/winograd.hip) for the 3x3 stride-1 layers with >= 256 channels on 16x16..64x64
maps (2.0-2.4x MIOpen), MIOpen otherwise (`select_wrw`).

`IPSR_CONV_ENGINE=miopen|direct|winograd|auto` (default auto) forces one engine wherever it is implemented — for the
per-engine parity tests and for A/B timing.  bf16 activations (BASELINE config 5): "bf16d" (csrc/conv_bf16.hip, the direct bf16
implicit GEMM, forward / input gradient / weight gradient of the k3 s1 p1 layers) and the split-bf16 Winograd engines where they win;
non-contiguous inputs and other dtypes take MIOpen.
"""
import functools
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import dist as ipsr_dist

_FORCE = None          # test hook: overrides the environment
# Arithmetic of the Winograd GEMMs (ops.MATH_CODE): "fp32" for fp32 activations (the reference's arithmetic; "bf16x6" / "bf16x3" are
# opt-in, models/IPSR.py `opt.conv_math`), "bf16x3" for bf16 activations / under bf16 autocast (BASELINE config 5).
_MATH = {"fp32": "fp32", "bf16": "bf16x3"}
_BF16_ENGINES = ("winograd", "wino_dil", "wino_s2", "bf16d")         # the engines that read / write bf16 activation tensors


# fp32 engines whose operands are small next to their weight stream / single pass (the innermost levels, netD's one-channel head): under
# bf16 activations they run on fp32 copies of the activations (a cast of a few hundred KB) instead of falling back to MIOpen's
# transposes + 40-160 us kernels
_CAST_ENGINES = ("one", "smallmap")


def set_conv_math(fp32=None, bf16=None):
    """Choose the arithmetic of the Winograd engines for fp32 activations and for bf16 activations (autocast)."""
    from .. import ops as _ops
    for key, val in (("fp32", fp32), ("bf16", bf16)):
        if val is not None:
            if val not in _ops.MATH_CODE:
                raise ValueError("conv math must be one of %s" % sorted(k for k in _ops.MATH_CODE if k))
            _MATH[key] = val


def _amp_bf16():
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
_check_hook = None     # test hook: callable(kind, engine, geometry, operands, result) after every engine call of _HipConv


# The A/B switches (IPSR_CONV_ENGINE, IPSR_NO_SMALLMAP, IPSR_NO_THIN, IPSR_BF16_ENGINES) are read from the environment ONCE, at first use:
# a training step asks `select` ~250 times, and each os.environ.get costs ~0.8 us of a host that is the step's bound under bf16
# (tools/profile_host.py: 2000 look-ups, 1.6 ms per step).  `reload_env()` re-reads them (a tool that flips one in-process).
_ENV = {}
_SEL = {}


def _env(name, default):
    v = _ENV.get(name)
    if v is None:
        v = _ENV[name] = os.environ.get(name, default)
    return v


def reload_env():
    _ENV.clear()
    _SEL.clear()
    for f in (_select, _select_wrw, _bf16_direct, _bf16_direct_wrw):
        f.cache_clear()


def _mode():
    return _FORCE or _env("IPSR_CONV_ENGINE", "auto")


def select(op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16=False):
    """-> the engine for one convolution call.  (Cin, H, W) = the module's input, as in ipsr_conv2d.  Memoised per (mode, shape):
    the rules query the library (workspace probes), ~150 convolution calls per training step ask.
    bf16: the activations are bf16 tensors — only the Winograd engines read / write those; every other shape takes MIOpen."""
    key = (0, _FORCE, op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16)
    eng = _SEL.get(key)
    if eng is None:
        eng = _SEL[key] = _select_any(op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16)
    return eng


def _select_any(op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16):
    eng = _select(_mode(), _env("IPSR_NO_SMALLMAP", "0") + _env("IPSR_NO_THIN", "0") + _env("IPSR_SMALLMAP_MAX_POS", "32"), op, B, Cin, H, W, Cout, k, stride, pad, dil)
    if not bf16:
        return eng
    if _bf16_wins(eng, Cin, H, W, Cout) or eng in _CAST_ENGINES:
        return eng
    if _mode() == "auto" and _thin_wins(op, B, Cin, H, W, Cout, k, stride, pad, dil, True):
        return "thin"            # the vector-ALU stream kernels read / write bf16 tensors themselves (ipsr_conv3x3_thin_io)
    if _mode() == "auto" and _env("IPSR_NO_THIN", "0") != "1" and op == ops.CONV_FWD and Cin == 3 and (k, stride, pad, dil) == (4, 2, 1, 1) and H * W >= 4096 \
            and ops.thin_f2m_mfma_supported(op, B, Cin, H, W, Cout, k, stride):
        # the first Conv2d of netP / netD (3 -> 64, k4 s2): the window gather on the matrix cores, 0.041-0.045 vs MIOpen's 0.059 ms and one
        # launch instead of four (profiles/r04_thin_bf16.txt; for the 3x3 thin layers the vector-ALU kernels and MIOpen stay ahead or level)
        return "thin_f2m"
    return _bf16_direct(_env("IPSR_BF16_ENGINES", ""), _mode(), op, B, Cin, H, W, Cout, k, stride, pad, dil)


@functools.lru_cache(maxsize=4096)
def _bf16_direct(force, mode, op, B, Cin, H, W, Cout, k, stride, pad, dil):
    """bf16 activations, and the split-bf16 Winograd engines do not win this shape: the DIRECT bf16 implicit GEMM (csrc/conv_bf16.hip,
    ops.conv3x3_bf16: one launch, NCHW in and out) where it is implemented — k3 s1 p1 on maps of 16..128 pixels width — else MIOpen.
    Measured at batch 16 (profiles/r04_conv_bf16_layers.txt): 700-870 TF against MIOpen's 350-570 incl. its layout transposes on
    every map from 32x32 up; on 16x16 maps MIOpen ties (and the Winograd engines win from 512 channels)."""
    if force == "none" or mode in ("miopen", "winograd", "direct"):
        return "miopen"
    if k == 3 and stride == 1 and pad == 1 and dil == 1 and H * W >= 256 and ops.conv3x3_bf16_supported(op, B, Cin, H, W, Cout):
        return "bf16d"           # 16x16 maps too since the reduction is cut over workgroups there (0.041-0.063 vs 0.056-0.081 split-Winograd, 0.070-0.122 MIOpen)
    # (measured and not kept: the small-map engine on fp32 copies for the innermost levels beyond 32 positions per batch — 708 images/s at
    # 32, 706 at 64, 696 at 256, 658 at 1024: MIOpen's tiny bf16 convolutions cost 45-60 us per CALL but far less device time)
    g = _s2_geometry(op in (ops.CONVT_FWD, ops.CONVT_BWD_DATA), B, Cin, H, W, Cout, k, stride, pad, dil)
    if g is not None and ops.conv4x4s2_bf16_supported(_s2_mode(op), B, *g):
        # the 4x4 stride-2 family (profiles/r04_conv_bf16_layers.txt, batch 16): 1.3-2x MIOpen wherever the launch has enough tiles;
        # a 16x16 coarse grid gives one pixel tile per image (64 workgroups at 512 channels) and MIOpen ties or wins
        Kc, Cf, nh, nw = g
        if _s2_mode(op) == ops.S2_FINE_TO_COARSE:
            if nw == 16:                 # one pixel tile per image: the split reduction fills the chip up to 512 produced channels (0.048-0.078 vs MIOpen's
                return "bf16d" if Kc <= 512 else "miopen"      # 0.083-0.106); 1024 produced channels leave no room to split: 0.098 vs 0.087
            return "bf16d" if nw >= 32 and ((Kc + 127) // 128) * B * nh * nw // 256 >= 128 else "miopen"
        return "bf16d"
    return "miopen"


def _bf16_wins(eng, Cin, H, W, Cout, wrw=False):
    """bf16 activations (BASELINE config 5).  MIOpen's bf16 implicit GEMMs run ~490 TF and need no transform passes; the Winograd
    engines here must SPLIT their operands (F(4x4,3x3) amplifies rounding ~100x), so their intermediates stay fp32-wide: 4.5x the
    bf16 activation bytes each way.  They win where the channel count amortises that (profiles/r03_bf16_layers.txt, batch 16, forward
    + both gradients against MIOpen incl. its layout transposes and weight casts): the 3x3 layers with >= 512 channels on one side at
    <= 32x32 (0.72-0.79x), the dilated 4x4 layers with >= 256 channels at <= 64x64 (0.70-0.90x), netD's 4x4 stride-1 layer (0.95x);
    they lose on larger maps and on the whole 4x4 stride-2 family (1.1-1.7x) — those stay on MIOpen.  IPSR_BF16_ENGINES=all|none
    overrides (A/B timing)."""
    force = _env("IPSR_BF16_ENGINES", "")
    if eng not in _BF16_ENGINES or force == "none":
        return False
    if force == "all":
        return True
    if eng == "winograd":
        # round 4: against the DIRECT bf16 kernel (csrc/conv_bf16.hip, profiles/r04_conv_bf16_layers.txt) the split engines keep the WEIGHT
        # GRADIENTS of the 16x16 maps (0.060-0.084 vs 0.076-0.159 ms); forward / input gradient went to the direct kernel when its reduction
        # was cut over workgroups; at 32x32 the direct weight gradient ties or wins since its runs were halved (0.068-0.131 vs 0.084-0.126)
        return wrw and H * W <= 256 and max(Cin, Cout) >= 512
    if eng == "wino_dil":
        return H * W <= 4096 and min(Cin, Cout) >= 256
    return False


@functools.lru_cache(maxsize=4096)
def _select(mode, _nosm, op, B, Cin, H, W, Cout, k, stride, pad, dil):
    fwd = op in (ops.CONV_FWD, ops.CONVT_FWD)
    cred, kout = (Cin, Cout) if fwd else (Cout, Cin)            # reduction / produced channels of this operation
    wino_ok = k == 3 and stride == 1 and pad == 1 and dil == 1 and cred % 16 == 0
    if mode == "miopen":
        return "miopen"
    if mode == "winograd":
        return "winograd" if wino_ok else "miopen"
    if mode == "direct":
        return "direct" if ops.conv2d_supported(op, B, Cin, H, W, Cout, k, stride, pad, dil) else "miopen"
    if mode == "auto" and _is_dilated4(k, stride, pad, dil) and op in (ops.CONV_FWD, ops.CONV_BWD_DATA) \
            and cred % 16 == 0 and min(Cin, Cout) >= 64 and 32 <= H <= 256 and H % 2 == 0 and W % 2 == 0:
        # netG's dilated down convolution: F(3x3,4x4), 1.4-2.0x MIOpen (profiles/r02_hipconv_k3_v3.txt); since the GEMM has a
        # 64-row tile also the outermost 64 -> 64 @256x256 level (0.171 / 0.193 vs 0.230 / 0.261 ms, profiles/r03_hipconv_k3.txt)
        return "wino_dil"
    if mode == "auto" and _is_k4s1(k, stride, pad, dil) and op in (ops.CONV_FWD, ops.CONV_BWD_DATA) \
            and cred % 16 == 0 and min(Cin, Cout) >= 128 and 16 <= H <= 128:
        return "wino_dil"        # netD's 4x4 stride-1 convolution: the same F(3x3,4x4) pipeline on the image itself
    if mode == "auto" and op == ops.CONV_FWD and Cout == 1 and Cin >= 64 and ops.conv_to_one_supported(B, Cin, H, W, k, stride, pad, dil):
        return "one"             # netD's last layer (512 -> 1): a single pass over the input, 15 vs 80-143 us
    if mode == "auto" and _thin_wins(op, B, Cin, H, W, Cout, k, stride, pad, dil):
        return "thin"            # 3/6-channel side at full resolution: one pass over the wide tensor on the vector ALUs
    if mode == "auto" and _smallmap_data_wins(op, B, Cin, H, W, Cout, k, stride, pad, dil):
        return "smallmap"        # innermost levels (<= 32 positions per batch): the weight tensor streamed once into MFMA operands
    if mode == "auto":
        g = _s2_geometry(op in (ops.CONVT_FWD, ops.CONVT_BWD_DATA), B, Cin, H, W, Cout, k, stride, pad, dil)
        if g is not None and _s2_wins(g, _s2_mode(op), B) and ops.s2_winograd_supported(_s2_mode(op), B, *g):
            return "wino_s2"     # 4x4 stride-2 layers: polyphase Winograd F(5x5,2x2)
    # auto: measured rules (MI355X, batch 8; profiles/r03_hipconv_k3.txt).  64 produced channels run on the GEMM's 64-row tile:
    # 64 -> 64 @256x256 (VGG conv1_2, forward and input gradient) 0.311 vs MIOpen's 0.378 ms
    if wino_ok and H * W >= 256 and min(cred, kout) >= 64 and (max(cred, kout) >= 128 or kout == 64):
        return "winograd"
    if k == 4 and stride == 2 and dil == 2 and op == ops.CONV_BWD_DATA and Cin >= 128 and 16 <= H <= 128:
        return "direct"          # dilated 4x4 stride-2 input gradient: 8-12 % faster than MIOpen's f3x2_dilation2 + transposes
    return "miopen"


def _is_dilated4(k, stride, pad, dil):
    return k == 4 and stride == 2 and pad == 3 and dil == 2


def _s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil):
    """(Kc, Cf, nh, nw) of a k4 s2 p1 layer in the coarse / fine terms of ipsr_conv4x4s2_winograd, or None."""
    if not (k == 4 and stride == 2 and pad == 1 and dil == 1):
        return None
    if transposed:
        return Cin, Cout, H, W
    if H % 2 or W % 2:
        return None
    return Cout, Cin, H // 2, W // 2


def _s2_wins(geom, mode=None, B=8):
    """Measured (profiles/r03_hipconv_k4s2.txt, batch 8): F(5x5,2x2) beats MIOpen by 5-35 % from 128 coarse / 64 fine channels up
    on coarse grids of 16..64; it loses on 8x8 and below (too few tiles).  The 64-channel 128x128 layer is transform bound: only
    its coarse-to-fine pass (ConvTranspose2d forward), whose output transform writes whole rows, is ahead (0.210 vs 0.226 ms)."""
    Kc, Cf, nh, nw = geom
    if mode == ops.S2_COARSE_TO_FINE and Kc >= 64 and Cf >= 64 and 16 <= min(nh, nw) and max(nh, nw) <= 128:
        return True
    if mode == ops.S2_COARSE_TO_FINE and B >= 16 and Kc >= 512 and Cf >= 512 and min(nh, nw) == 8 and max(nh, nw) == 8:
        return True          # netF's 512 -> 512 @16 -> 8 input gradient at batch 16 (the batched discriminator pass): 0.141 vs 0.188 ms
    return Kc >= 128 and Cf >= 64 and 16 <= min(nh, nw) and max(nh, nw) <= 64


def _s2_mode(op):
    return ops.S2_FINE_TO_COARSE if op in (ops.CONV_FWD, ops.CONVT_BWD_DATA) else ops.S2_COARSE_TO_FINE


def _smallmap_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil):
    """(B, R, Cq, Ho, Wo, Hf, Wf, k, stride, pad, dil) of ipsr_conv_smallmap for this module call: R / (Ho, Wo) = the weight's first
    channel dimension and its grid, Cq / (Hf, Wf) = the second."""
    if transposed:
        Hy, Wy = (H - 1) * stride - 2 * pad + dil * (k - 1) + 1, (W - 1) * stride - 2 * pad + dil * (k - 1) + 1
        return B, Cin, Cout, H, W, Hy, Wy, k, stride, pad, dil
    Hy, Wy = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    return B, Cout, Cin, Hy, Wy, H, W, k, stride, pad, dil


def _thin_wins(op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16=False):
    """3x3 stride-1 layers with a 3- or 6-channel side on maps of >= 64x64 (profiles/r02_thin.txt, device time at 256x256, batch 8):
    many -> few (VGG conv1_1 input gradient 185 -> 61 us, netG's last ConvTranspose2d forward 217 -> 130 us) and 3 -> many
    (VGG conv1_1 forward 70 -> 52 us, and the bias + ReLU pass goes into the kernel); 6 -> 64 forward and the weight gradients
    stay on MIOpen (87 vs 99 us; 134 vs 296 us).  bf16 activations (batch 16, profiles/r04_thin_bf16.txt): 3 -> many wins by 2x and
    1.5x (VGG conv1_1 forward 90 vs 182 us, bias + ReLU included; the last ConvTranspose2d's input gradient 172 vs 261 us); 6 -> 64 stays
    on MIOpen (175 vs 156 us) and many -> 3 on the direct MFMA kernel (252 vs 300 us)."""
    if _env("IPSR_NO_THIN", "0") == "1":           # A/B switch
        return False
    if not (k == 3 and stride == 1 and pad == 1 and dil == 1) or H * W < 4096 or not ops.thin_supported(op, Cin, H, W, Cout):
        return False
    fwd = op in (ops.CONV_FWD, ops.CONVT_FWD)
    i, o = (Cin, Cout) if fwd else (Cout, Cin)
    if bf16:
        return i == 3
    return o in (3, 6) or i == 3


def _smallmap_op(op):
    """Conv2d forward / ConvTranspose2d backward-data contract the weight's second dimension (SM_FWD); the other two its first."""
    return ops.SM_FWD if op in (ops.CONV_FWD, ops.CONVT_BWD_DATA) else ops.SM_DATA


def _smallmap_data_wins(op, B, Cin, H, W, Cout, k, stride, pad, dil):
    """Forward / input gradient on grids of <= 32 positions per batch (2x2 and 1x1 at batch 8): 15-20 us on the device against
    MIOpen's 40-50 (profiles/r02_hipconv_small.txt); from 128 positions up the op is a real GEMM and MIOpen ties."""
    if _env("IPSR_NO_SMALLMAP", "0") == "1" or k not in (3, 4) or min(Cin, Cout) < 256:
        return False
    g = _smallmap_geometry(op in (ops.CONVT_FWD, ops.CONVT_BWD_DATA), B, Cin, H, W, Cout, k, stride, pad, dil)
    return g[3] >= 1 and g[4] >= 1 and g[0] * g[3] * g[4] <= int(_env("IPSR_SMALLMAP_MAX_POS", "32")) and ops.smallmap_supported(_smallmap_op(op), *g)


def _smallmap_wrw_wins(transposed, B, Cin, H, W, Cout, k, stride, pad, dil):
    """Weight gradients of the 4x4 layers on grids of <= 256 positions per batch (the four innermost levels at batch 8): the GEMM
    writes dW in place, 27-39 us against MIOpen's 41-57 (profiles/r02_hipconv_small.txt)."""
    if k != 4 or stride != 2 or _env("IPSR_NO_SMALLMAP", "0") == "1":        # the switch is for A/B timing
        return False
    g = _smallmap_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil)
    return g[3] >= 1 and g[4] >= 1 and g[0] * g[3] * g[4] <= 256 and min(Cin, Cout) >= 256 and ops.smallmap_supported(ops.SM_WRW, *g)


def _is_k4s1(k, stride, pad, dil):
    return k == 4 and stride == 1 and pad == 1 and dil == 1


def select_wrw(transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16=False):
    """-> the engine for the weight gradient of one layer (profiles/r02_hipconv_k3_wrw.txt: F(3x3,4x4) is 2.0-2.4x
    MIOpen from 256 channels up on maps of 16x16..64x64; on larger maps its tile-major transforms lose to MIOpen)."""
    key = (1, _FORCE, transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16)
    eng = _SEL.get(key)
    if eng is None:
        eng = _SEL[key] = _select_wrw_any(transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16)
    return eng


def _select_wrw_any(transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16):
    eng = _select_wrw(_mode(), _env("IPSR_NO_SMALLMAP", "0"), transposed, B, Cin, H, W, Cout, k, stride, pad, dil)
    thin = _mode() == "auto" and _env("IPSR_NO_THIN", "0") != "1" and pad == 1 and dil == 1 and (k, stride) in ((3, 1), (4, 2)) and H * W >= 4096 \
        and ops.thin_wrw_mfma_supported(transposed, B, Cin, H, W, Cout, k, stride)
    if not bf16:
        # fp32 activations: the same pixel reduction on v_mfma_f32_32x32x2_f32 (profiles/r04_thin_fp32.txt, batch 8)
        return "thin_mfma" if thin and eng == "miopen" else eng
    if _bf16_wins(eng, Cin, H, W, Cout, True) or eng in _CAST_ENGINES:
        return eng
    if thin:
        # 3 / 6 channels on the narrow side: the pixel reduction on the bf16 matrix cores straight from NCHW (profiles/r04_thin_bf16.txt, batch 16:
        # 3 -> 64 k4 s2 0.033 vs MIOpen's 0.057 ms, ConvT 128 -> 3 k3 0.152 vs 0.202, k4 s2 0.052 vs 0.063, 6 -> 64 0.118 vs 0.125 — and 2
        # launches instead of MIOpen's 5-6)
        return "thin_mfma"
    return _bf16_direct_wrw(_env("IPSR_BF16_ENGINES", ""), _mode(), transposed, B, Cin, H, W, Cout, k, stride, pad, dil)


@functools.lru_cache(maxsize=4096)
def _bf16_direct_wrw(force, mode, transposed, B, Cin, H, W, Cout, k, stride, pad, dil):
    """Weight gradient on bf16 activations by the direct kernel (ops.conv3x3_bf16_wrw): 1.1-2.0x MIOpen on every map from 32x32 up (one
    run per CU: the partial-sum slabs are what it costs)."""
    if force == "none" or mode in ("miopen", "winograd", "direct"):
        return "miopen"
    if k == 3 and stride == 1 and pad == 1 and dil == 1 and H * W >= 1024 and ops.conv3x3_bf16_wrw_supported(transposed, B, Cin, H, W, Cout):
        return "bf16d"
    g = _s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil)
    if g is not None and g[3] >= 32 and ((g[0] + 127) // 128) * ((g[1] + 31) // 32) >= 4 and ops.conv4x4s2_bf16_wrw_supported(B, *g):
        return "bf16d"           # 1.2-1.4x MIOpen from four 128 x 32 output tiles up on coarse grids >= 32 wide; 16-wide grids and single tiles lose
    return "miopen"


@functools.lru_cache(maxsize=4096)
def _select_wrw(mode, _nosm, transposed, B, Cin, H, W, Cout, k, stride, pad, dil):
    if mode == "auto" and not transposed and Cout == 1 and Cin >= 64 and ops.conv_to_one_supported(B, Cin, H, W, k, stride, pad, dil):
        return "one"
    if mode == "auto" and not transposed and _is_dilated4(k, stride, pad, dil) and min(Cin, Cout) >= 128 and 32 <= H <= 128 \
            and H % 2 == 0 and W % 2 == 0:
        return "wino_dil"
    if mode == "auto" and not transposed and _is_k4s1(k, stride, pad, dil) and min(Cin, Cout) >= 128 and 16 <= H <= 128:
        return "wino_dil"
    if mode == "auto":
        g = _s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil)
        if g is not None and _s2_wins(g):
            return "wino_s2"
    if mode == "auto" and _smallmap_wrw_wins(transposed, B, Cin, H, W, Cout, k, stride, pad, dil):
        return "smallmap"
    ok = k == 3 and stride == 1 and pad == 1 and dil == 1
    if mode in ("miopen", "direct") or not ok:
        return "miopen"
    if mode == "winograd":
        return "winograd"
    if 256 <= H * W <= 4096 and max(Cin, Cout) >= 256 and min(Cin, Cout) >= 128:
        return "winograd"
    if transposed and Cout == 64 and Cin >= 256 and H * W <= 16384:
        return "winograd"        # netG upconv_1 (256 -> 64 @128x128): the one large-map weight gradient that is ahead, 0.365 vs 0.409 ms
    return "miopen"


def _bf16_direct_call(op, inp, w, transposed, B, Cin, H, W, Cout, k, stride, pad, dil, out_dtype):
    """One pass of a module on the direct bf16 kernels (csrc/conv_bf16.hip): k3 s1 p1, or k4 s2 p1 in its coarse / fine form."""
    if k == 3:
        return ops.conv3x3_bf16(op, inp, w, (B, Cin, H, W), Cout, out_dtype=out_dtype, keep_packed=not w.requires_grad and not torch.is_grad_enabled())
    return ops.conv4x4s2_bf16(_s2_mode(op), inp, w, B, *_s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil), out_dtype=out_dtype)


def _miopen_forward(x, w, transposed, stride, pad, dil):
    if x.dtype != w.dtype and not torch.is_autocast_enabled():
        w = w.to(x.dtype)
    return F.conv_transpose2d(x, w, None, stride, pad, 0, 1, dil) if transposed else F.conv2d(x, w, None, stride, pad, dil)


def _miopen_backward(dy, x, w, transposed, stride, pad, dil, which):
    """aten.convolution_backward on operands of ONE dtype (bf16 activations: the fp32 weight is cast, its gradient cast back)."""
    dt = dy.dtype
    out = torch.ops.aten.convolution_backward(dy, x if x.dtype == dt else x.to(dt), w if w.dtype == dt else w.to(dt), None, [stride, stride],
                                              [pad, pad], [dil, dil], transposed, [0, 0], 1, which)
    return out[0], (out[1].to(w.dtype) if out[1] is not None else None)


class _HipConv(torch.autograd.Function):
    """Bias-free Conv2d / ConvTranspose2d: forward, input gradient and weight gradient each on the engine `select` names for it
    (MIOpen where none is faster).  `math` = the Winograd engines' arithmetic, `act` = dtype of the activation tensors produced."""

    @staticmethod
    def forward(ctx, x, w, transposed, k, stride, pad, dil, eng_fwd, math, act):
        B, Cin, H, W = x.shape
        Cout = w.shape[1] if transposed else w.shape[0]
        op = ops.CONVT_FWD if transposed else ops.CONV_FWD
        xc = x.contiguous()
        if eng_fwd == "winograd":
            y = ops.conv3x3_winograd(op, xc, w, (B, Cin, H, W), Cout, math=math, out_dtype=act)
        elif eng_fwd == "direct":
            y = ops.conv2d(op, xc, w, (B, Cin, H, W), Cout, k, stride, pad, dil)
        elif eng_fwd == "bf16d":
            y = _bf16_direct_call(op, xc if xc.dtype == torch.bfloat16 else xc.to(torch.bfloat16), w, transposed, B, Cin, H, W, Cout, k, stride, pad, dil, act)
        elif eng_fwd == "wino_dil":
            y = ops.conv4x4_dilated_winograd(0, xc, w, (B, Cin, H, W), Cout, geom=ops.conv4x4_geometry(k, stride, pad, dil), math=math, out_dtype=act)
        elif eng_fwd == "thin":
            y = ops.conv3x3_thin(op, xc, w, (B, Cin, H, W), Cout, out_dtype=act)
        elif eng_fwd == "thin_f2m":
            y = ops.conv_thin_f2m_mfma(op, xc, w, (B, Cin, H, W), Cout, k, stride, out_dtype=act)
        elif eng_fwd == "one":
            y = ops.conv_to_one(xc.float(), w, pad).to(act)
        elif eng_fwd == "smallmap":
            y = ops.conv_smallmap(_smallmap_op(op), xc.float(), w, *_smallmap_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil)).to(act)
        elif eng_fwd == "wino_s2":
            y = ops.conv4x4s2_winograd(_s2_mode(op), xc, w, B, *_s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil), math=math, out_dtype=act)
        else:
            y = _miopen_forward(xc, w, transposed, stride, pad, dil)
        ctx.save_for_backward(xc, w)
        ctx.geom = (transposed, k, stride, pad, dil, Cout)
        ctx.math, ctx.bf16 = math, act == torch.bfloat16
        if _check_hook is not None:
            _check_hook("forward", eng_fwd, ctx.geom, (xc, w), y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        transposed, k, stride, pad, dil, Cout = ctx.geom
        math, bf16 = ctx.math, ctx.bf16
        B, Cin, H, W = x.shape
        dy = dy.contiguous()
        if bf16 and dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            op = ops.CONVT_BWD_DATA if transposed else ops.CONV_BWD_DATA
            eng = select(op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16)
            if eng == "winograd":
                dx = ops.conv3x3_winograd(op, dy, w, (B, Cin, H, W), Cout, math=math, out_dtype=x.dtype)
            elif eng == "direct":
                dx = ops.conv2d(op, dy, w, (B, Cin, H, W), Cout, k, stride, pad, dil)
            elif eng == "bf16d":
                dx = _bf16_direct_call(op, dy, w, transposed, B, Cin, H, W, Cout, k, stride, pad, dil, x.dtype)
            elif eng == "wino_dil":
                dx = ops.conv4x4_dilated_winograd(1, dy, w, (B, Cin, H, W), Cout, geom=ops.conv4x4_geometry(k, stride, pad, dil), math=math, out_dtype=x.dtype)
            elif eng == "thin":
                dx = ops.conv3x3_thin(op, dy, w, (B, Cin, H, W), Cout, out_dtype=x.dtype)
            elif eng == "smallmap":
                dx = ops.conv_smallmap(_smallmap_op(op), dy.float(), w, *_smallmap_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil)).to(x.dtype)
            elif eng == "wino_s2":
                dx = ops.conv4x4s2_winograd(_s2_mode(op), dy, w, B, *_s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil), math=math, out_dtype=x.dtype)
            else:
                dx = _miopen_backward(dy, x, w, transposed, stride, pad, dil, [True, False, False])[0]
                if dx.dtype != x.dtype:
                    dx = dx.to(x.dtype)
            if _check_hook is not None:
                _check_hook("input_grad", eng, ctx.geom, (dy, x, w), dx)
        weng = select_wrw(transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16) if ctx.needs_input_grad[1] else None
        # data parallel: write the weight gradient straight into its slice of the armed gradient bucket (dist.py)
        sink = ipsr_dist.grad_sink_for(w.data_ptr(), w.shape) if weng in ("winograd", "wino_dil", "wino_s2", "smallmap", "one", "bf16d", "thin_mfma") else None
        xw = x if (x.dtype == dy.dtype or weng in (None, "miopen", "thin_mfma")) else x.to(dy.dtype)      # a weight gradient reads both operands in one dtype
        if weng == "winograd":
            dw = ops.conv3x3_winograd_wrw(transposed, xw, dy, Cout, out=sink, math=math)
        elif weng == "thin_mfma":
            xa = x.to(torch.bfloat16) if (bf16 and transposed and x.dtype != torch.bfloat16) else x       # the WIDE tensor decides the arithmetic
            dw = ops.conv_thin_wrw_mfma(transposed, xa, dy, k, stride, out=sink)
        elif weng == "bf16d" and k == 3:
            dw = ops.conv3x3_bf16_wrw(transposed, xw, dy, Cout, out=sink)
        elif weng == "bf16d":
            fine, coarse = (dy, xw) if transposed else (xw, dy)
            dw = ops.conv4x4s2_bf16_wrw(fine, coarse, B, *_s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil), out=sink)
        elif weng == "wino_dil":
            dw = ops.conv4x4_dilated_winograd(2, xw, dy, (B, Cin, H, W), Cout, out=sink, geom=ops.conv4x4_geometry(k, stride, pad, dil), math=math)
        elif weng == "one":
            dw = ops.conv_to_one_wrw(x.float(), dy.float(), k, pad, out=sink)
        elif weng == "smallmap":
            coarse, fine = (x.float(), dy.float()) if transposed else (dy.float(), x.float())
            dw = ops.conv_smallmap(ops.SM_WRW, coarse, fine, *_smallmap_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil), out=sink)
        elif weng == "wino_s2":
            fine, coarse = (dy, xw) if transposed else (xw, dy)
            dw = ops.conv4x4s2_winograd(ops.S2_WEIGHT_GRAD, fine, coarse, B, *_s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil), out=sink, math=math)
        elif ctx.needs_input_grad[1]:
            dw = _miopen_backward(dy, x, w, transposed, stride, pad, dil, [False, True, False])[1]
        if _check_hook is not None and dw is not None:
            _check_hook("weight_grad", weng or "miopen", ctx.geom, (dy, x, w), dw)
        return dx, dw, None, None, None, None, None, None, None, None


def _geometry(m):
    """(k, stride, pad, dil) of a module the kernels can express (square, symmetric, groups 1, no output padding) or None."""
    ks, st, pd, dl = m.kernel_size, m.stride, m.padding, m.dilation
    if m.groups != 1 or ks[0] != ks[1] or st[0] != st[1] or pd[0] != pd[1] or dl[0] != dl[1] or isinstance(pd, str):
        return None
    if isinstance(m, nn.ConvTranspose2d) and tuple(m.output_padding) != (0, 0):
        return None
    if getattr(m, "padding_mode", "zeros") != "zeros":
        return None
    return ks[0], st[0], pd[0], dl[0]


def any_engine(m, x):
    """True when at least one pass of m(x) would run on a HIP engine (so the caller should split the bias off and come through
    conv_nobias); False = all three passes are MIOpen's and the plain module call loses nothing."""
    g = _geometry(m)
    if g is None or not x.is_cuda or x.dim() != 4 or m.weight.dtype != torch.float32 or x.dtype not in (torch.float32, torch.bfloat16) \
            or (torch.is_autocast_enabled() and not _amp_bf16()):
        return False
    transposed = isinstance(m, nn.ConvTranspose2d)
    k, stride, pad, dil = g
    B, Cin, H, W = x.shape
    Cout = m.weight.shape[1] if transposed else m.weight.shape[0]
    bf16 = _amp_bf16() or x.dtype == torch.bfloat16
    fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if transposed else (ops.CONV_FWD, ops.CONV_BWD_DATA)
    return select(fop, B, Cin, H, W, Cout, k, stride, pad, dil, bf16) != "miopen" \
        or select(bop, B, Cin, H, W, Cout, k, stride, pad, dil, bf16) != "miopen" \
        or select_wrw(transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16) != "miopen"


def conv_nobias(m, x, weight=None):
    """m(x) without the bias (the fused epilogue kernels add it): HIP engine where `select` says so, else MIOpen.  fp32 activations:
    every engine; bf16 activations (a bf16 tensor, or any input under bf16 autocast — BASELINE config 5): the Winograd engines, which
    read / write bf16 and multiply split-bf16 operands on the bf16 matrix cores (`_MATH`); the weights stay fp32 parameters."""
    w = m.weight if weight is None else weight
    transposed = isinstance(m, nn.ConvTranspose2d)
    g = _geometry(m)
    amp = _amp_bf16()
    if g is not None and x.is_cuda and w.dtype == torch.float32 and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) \
            and (amp or not torch.is_autocast_enabled()):
        k, stride, pad, dil = g
        B, Cin, H, W = x.shape
        Cout = w.shape[1] if transposed else w.shape[0]
        op = ops.CONVT_FWD if transposed else ops.CONV_FWD
        bf16 = amp or x.dtype == torch.bfloat16
        act = torch.bfloat16 if bf16 else torch.float32
        math = _MATH["bf16" if bf16 else "fp32"]
        eng = select(op, B, Cin, H, W, Cout, k, stride, pad, dil, bf16)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
        if needs_grad:
            # the backward may use a HIP engine even where the forward stays on MIOpen
            bop = ops.CONVT_BWD_DATA if transposed else ops.CONV_BWD_DATA
            beng = select(bop, B, Cin, H, W, Cout, k, stride, pad, dil, bf16) if x.requires_grad else "miopen"
            weng = select_wrw(transposed, B, Cin, H, W, Cout, k, stride, pad, dil, bf16) if w.requires_grad else "miopen"
            if eng != "miopen" or beng != "miopen" or weng != "miopen":
                return _HipConv.apply(x, w, transposed, k, stride, pad, dil, eng, math, act)
        elif eng == "winograd":
            return ops.conv3x3_winograd(op, x.contiguous(), w.detach(), (B, Cin, H, W), Cout, math=math, out_dtype=act)
        elif eng == "direct":
            return ops.conv2d(op, x.contiguous(), w.detach(), (B, Cin, H, W), Cout, k, stride, pad, dil)
        elif eng == "bf16d":
            xb = x.contiguous()
            return _bf16_direct_call(op, xb if xb.dtype == torch.bfloat16 else xb.to(torch.bfloat16), w.detach(), transposed, B, Cin, H, W, Cout, k, stride, pad, dil, act)
        elif eng == "wino_dil":
            return ops.conv4x4_dilated_winograd(0, x.contiguous(), w.detach(), (B, Cin, H, W), Cout, geom=ops.conv4x4_geometry(k, stride, pad, dil),
                                                math=math, out_dtype=act)
        elif eng == "thin":
            return ops.conv3x3_thin(op, x.contiguous(), w.detach(), (B, Cin, H, W), Cout, out_dtype=act)
        elif eng == "thin_f2m":
            return ops.conv_thin_f2m_mfma(op, x.contiguous(), w.detach(), (B, Cin, H, W), Cout, k, stride, out_dtype=act)
        elif eng == "one":
            return ops.conv_to_one(x.contiguous().float(), w.detach(), pad).to(act)
        elif eng == "smallmap":
            return ops.conv_smallmap(_smallmap_op(op), x.contiguous().float(), w.detach(), *_smallmap_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil)).to(act)
        elif eng == "wino_s2":
            return ops.conv4x4s2_winograd(_s2_mode(op), x.contiguous(), w.detach(), B, *_s2_geometry(transposed, B, Cin, H, W, Cout, k, stride, pad, dil),
                                          math=math, out_dtype=act)
    if transposed:
        return F.conv_transpose2d(x, w, None, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
    return F.conv2d(x, w, None, m.stride, m.padding, m.dilation, m.groups)
