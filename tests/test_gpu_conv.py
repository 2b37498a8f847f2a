"""SURVEY §8 (f)1 — the hand-written convolution kernels (csrc/conv_gemm.hip implicit GEMM, csrc/winograd.hip F(4x4,3x3))
against an fp64 torch convolution, per geometry of the reference's nets (models/networks.py:220-259, 404-432, 470-495,
510-515; models/vgg16.py:9-21), through the C-ABI (ops.conv2d / ops.conv3x3_winograd) and through the module path
(models/hipconv.py: forward, input gradient, weight gradient)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (kind, Cin, H, W, Cout, k, stride, pad, dil): every geometry of the four nets + VGG, at sizes an fp64 CPU convolution
# finishes in well under a second; odd extents, channel counts that are not tile multiples, 1x1 maps, batch tails
GEOMS = [
    ("conv", 32, 16, 16, 48, 3, 1, 1, 1),          # downconv_3 / VGG
    ("conv", 130, 9, 13, 70, 3, 1, 1, 1),          # ragged: Cin not a multiple of 16 (direct only), odd map
    ("conv", 16, 20, 20, 24, 4, 2, 1, 1),          # netP / netD / netF down convolution
    ("conv", 16, 21, 19, 24, 4, 2, 1, 1),          # odd input extents
    ("conv", 24, 16, 16, 24, 4, 2, 3, 2),          # netG dilated down convolution
    ("conv", 24, 6, 6, 24, 4, 2, 3, 2),
    ("conv", 24, 2, 2, 24, 4, 2, 3, 2),            # innermost level
    ("conv", 40, 12, 12, 36, 4, 1, 1, 1),          # netD stride-1 4x4 (32 -> 31)
    ("conv", 256, 9, 9, 1, 4, 1, 1, 1),            # netD's last convolution: one output channel
    ("conv", 512, 2, 2, 512, 4, 2, 1, 1),          # netP innermost: 2x2 -> 1x1, long reduction (split-K)
    ("convT", 32, 16, 16, 48, 3, 1, 1, 1),         # upconv_3
    ("convT", 64, 5, 7, 20, 3, 1, 1, 1),
    ("convT", 16, 10, 10, 24, 4, 2, 1, 1),         # netP / netG up convolution
    ("convT", 16, 7, 9, 24, 4, 2, 1, 1),
    ("convT", 512, 1, 1, 64, 4, 2, 1, 1),          # innermost: 1x1 -> 2x2
]


def _ref64(kind, x, w, dy, st, pad, dil):
    """fp64 CPU forward and input gradient."""
    xd = x.double().cpu().requires_grad_(True)
    wd = w.double().cpu().requires_grad_(True)
    y = F.conv_transpose2d(xd, wd, None, st, pad, 0, 1, dil) if kind == "convT" else F.conv2d(xd, wd, None, st, pad, dil)
    dx, dw = torch.autograd.grad(y, (xd, wd), dy.double().cpu())
    return y.detach(), dx, dw


def _rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "%s_c%d_%dx%d_k%d_%ds%dp%dd%d" % (g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8]))
def test_direct_implicit_gemm_vs_fp64(geom):
    """conv_gemm.hip: forward and backward-data of every geometry within 2e-5 of fp64 (fp32 accumulation of <= 8192 terms)."""
    from deepinpainting_amd import ops
    kind, Cin, H, W, Cout, k, st, pad, dil = geom
    tr = kind == "convT"
    B = 3
    g = torch.Generator().manual_seed(Cin * 1000 + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), generator=g) * 0.1).cuda()
    Ho, Wo = ops.conv_out_dim(ops.CONVT_FWD if tr else ops.CONV_FWD, H, k, st, pad, dil), ops.conv_out_dim(ops.CONVT_FWD if tr else ops.CONV_FWD, W, k, st, pad, dil)
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).cuda()
    y64, dx64, _ = _ref64(kind, x, w, dy, st, pad, dil)
    fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
    assert ops.conv2d_supported(fop, B, Cin, H, W, Cout, k, st, pad, dil) and ops.conv2d_supported(bop, B, Cin, H, W, Cout, k, st, pad, dil)
    y = ops.conv2d(fop, x, w, (B, Cin, H, W), Cout, k, st, pad, dil)
    dx = ops.conv2d(bop, dy, w, (B, Cin, H, W), Cout, k, st, pad, dil)
    torch.cuda.synchronize()
    assert tuple(y.shape) == tuple(y64.shape) and tuple(dx.shape) == tuple(dx64.shape)
    assert _rel(y, y64) <= 2e-5
    assert _rel(dx, dx64) <= 2e-5
    # deterministic: the same call gives the same bits (fixed summation order, split-K reduced in order)
    assert torch.equal(y, ops.conv2d(fop, x, w, (B, Cin, H, W), Cout, k, st, pad, dil))


@pytest.mark.parametrize("kind,Cin,H,W,Cout", [("conv", 32, 16, 16, 48), ("conv", 128, 32, 32, 128), ("conv", 64, 9, 13, 48),
                                               ("conv", 512, 16, 16, 256), ("convT", 32, 16, 16, 48), ("convT", 256, 12, 20, 64),
                                               ("conv", 16, 4, 4, 16), ("conv", 48, 6, 10, 144)])
def test_winograd_f4x4_3x3_vs_fp64(kind, Cin, H, W, Cout):
    """winograd.hip: k3 s1 p1 forward and backward-data, Conv2d and ConvTranspose2d, within 1e-4 of fp64 — the transforms
    amplify rounding (coefficients up to 8, 1/24): measured 1e-5..3e-5 of the output scale; extents that are not multiples
    of 4 and tile counts that are not multiples of 128 included."""
    from deepinpainting_amd import ops
    tr = kind == "convT"
    B = 2
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3), generator=g) * 0.1).cuda()
    dy = torch.randn(B, Cout, H, W, generator=g).cuda()
    y64, dx64, _ = _ref64(kind, x, w, dy, 1, 1, 1)
    fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
    y = ops.conv3x3_winograd(fop, x, w, (B, Cin, H, W), Cout)
    dx = ops.conv3x3_winograd(bop, dy, w, (B, Cin, H, W), Cout)
    torch.cuda.synchronize()
    assert _rel(y, y64) <= 1e-4
    assert _rel(dx, dx64) <= 1e-4
    assert not ops.winograd_supported(fop, B, 24, H, W, Cout)          # reduction channels must be a multiple of 16: refuses


@pytest.mark.parametrize("Cin,H,W,Cout,B", [(64, 16, 16, 96, 2), (1024, 16, 16, 512, 8), (32, 6, 10, 40, 3)])
def test_winograd_fused_epilogues_and_filter_cache(Cin, H, W, Cout, B):
    """ipsr_conv3x3_winograd_ex: bias + ReLU and bias + ReLU + 2x2 max-pool in the output transform (the VGG16 chain), the
    cached filter transform, and the split reduction (the 1024 -> 512 @ 16x16 case splits 3 ways): equal to the plain call
    followed by torch's bias / relu / max_pool2d, bit for bit (same GEMM, same sums)."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    plain = ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, W), Cout)
    y64 = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
    assert _rel(plain, y64) <= 1e-4
    want_relu = torch.relu(plain + bias.view(1, -1, 1, 1))
    cache = ops.winograd_filter_cache(ops.CONV_FWD, Cin, Cout, x.device)
    got = ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, W), Cout, bias=bias, epilogue="relu", filter_cache=cache)
    assert torch.equal(got, want_relu)
    got2 = ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, W), Cout, bias=bias, epilogue="relu_pool", filter_cache=cache, filter_cache_valid=True)
    assert torch.equal(got2, F.max_pool2d(want_relu, 2, 2))
    # a stale cache is really used (proves the cached path skips the filter transform): zero it -> zero convolution
    cache.zero_()
    z = ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, W), Cout, bias=None, epilogue=None, filter_cache=cache, filter_cache_valid=True)
    assert not z.any()


@pytest.mark.parametrize("kind,Cin,H,W,Cout", [("conv", 32, 16, 16, 48), ("conv", 128, 32, 32, 160), ("conv", 20, 9, 13, 7),
                                               ("convT", 64, 16, 16, 24), ("convT", 256, 12, 20, 64), ("conv", 16, 4, 4, 16)])
def test_winograd_weight_gradient_f3x3_4x4_vs_fp64(kind, Cin, H, W, Cout):
    """winograd.hip, weight gradient of the k3 s1 p1 layers by F(3x3,4x4): within 1e-4 of an fp64 autograd gradient (any
    channel counts — the GEMM tiles are zero padded — and ragged extents)."""
    from deepinpainting_amd import ops
    tr = kind == "convT"
    B = 3
    g = torch.Generator().manual_seed(Cin * 7 + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3), generator=g) * 0.1).cuda()
    dy = torch.randn(B, Cout, H, W, generator=g).cuda()
    _, _, dw64 = _ref64(kind, x, w, dy, 1, 1, 1)
    dw = ops.conv3x3_winograd_wrw(tr, x, dy, Cout)
    torch.cuda.synchronize()
    assert tuple(dw.shape) == tuple(dw64.shape)
    assert _rel(dw, dw64) <= 1e-4


@pytest.mark.parametrize("Cin,H,W,Cout,B", [(32, 16, 16, 48, 2), (128, 32, 32, 128, 3), (16, 6, 10, 24, 2), (64, 2, 2, 64, 8), (48, 14, 22, 20, 1)])
def test_dilated_4x4_winograd_f3x3_4x4_vs_fp64(Cin, H, W, Cout, B):
    """netG's Conv2d(k4, stride 2, pad 3, dilation 2) (models/networks.py:226) by Winograd F(3x3,4x4): forward, input gradient
    (odd rows / columns only — the even ones are exactly zero, as in the reference's autograd) and weight gradient within
    1e-4 of fp64, on output grids that are and are not multiples of the 3x3 tile."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Cin * 3 + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1).cuda()
    dy = torch.randn(B, Cout, H // 2, W // 2, generator=g).cuda()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    y64 = F.conv2d(xd, wd, None, 2, 3, 2)
    dx64, dw64 = torch.autograd.grad(y64, (xd, wd), dy.double().cpu())
    y = ops.conv4x4_dilated_winograd(0, x, w, (B, Cin, H, W), Cout)
    dw = ops.conv4x4_dilated_winograd(2, x, dy, (B, Cin, H, W), Cout)
    assert _rel(y, y64.detach()) <= 1e-4
    assert _rel(dw, dw64) <= 1e-4
    if Cout % 16 == 0:
        dx = ops.conv4x4_dilated_winograd(1, dy, w, (B, Cin, H, W), Cout)
        assert _rel(dx, dx64) <= 1e-4
        assert not dx[:, :, 0::2, :].any() and not dx[:, :, :, 0::2].any()
    else:
        assert not ops.dilated_winograd_supported(1, B, Cin, H, W, Cout)


@pytest.mark.parametrize("Cin,H,W,Cout,B", [(32, 16, 16, 48, 2), (256, 32, 32, 128, 2), (16, 7, 10, 32, 3), (64, 4, 4, 64, 4), (48, 13, 21, 20, 1)])
def test_k4_s1_p1_winograd_f3x3_4x4_vs_fp64(Cin, H, W, Cout, B):
    """netD's fourth convolution Conv2d(k4, stride 1, pad 1) (models/networks.py:483-489; 256 -> 512 on 32x32 -> 31x31) through
    the same F(3x3,4x4) pipeline (geometry 1 of ipsr_conv4x4_winograd): forward, input gradient (every position) and weight
    gradient within 1e-4 of fp64, on odd and even extents and grids that are not multiples of the 3x3 tile."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Cin * 5 + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * 0.1).cuda()
    dy = torch.randn(B, Cout, H - 1, W - 1, generator=g).cuda()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    y64 = F.conv2d(xd, wd, None, 1, 1, 1)
    dx64, dw64 = torch.autograd.grad(y64, (xd, wd), dy.double().cpu())
    G1 = ops.GEOM_K4_S1_P1
    assert ops.conv4x4_geometry(4, 1, 1, 1) == G1 and ops.conv4x4_geometry(4, 2, 3, 2) == ops.GEOM_K4_S2_P3_D2 and ops.conv4x4_geometry(4, 2, 1, 1) is None
    y = ops.conv4x4_dilated_winograd(0, x, w, (B, Cin, H, W), Cout, geom=G1)
    dw = ops.conv4x4_dilated_winograd(2, x, dy, (B, Cin, H, W), Cout, geom=G1)
    assert tuple(y.shape) == (B, Cout, H - 1, W - 1)
    assert _rel(y, y64.detach()) <= 1e-4
    assert _rel(dw, dw64) <= 1e-4
    if Cout % 16 == 0:
        dx = torch.full((B, Cin, H, W), float("nan"), device="cuda")          # every element must be overwritten
        ops.conv4x4_dilated_winograd(1, dy, w, (B, Cin, H, W), Cout, out=dx, geom=G1)
        assert _rel(dx, dx64) <= 1e-4
    else:
        assert not ops.dilated_winograd_supported(1, B, Cin, H, W, Cout, geom=G1)


@pytest.mark.parametrize("Kc,Cf,nh,nw,B", [(32, 16, 8, 8, 2), (128, 64, 16, 16, 2), (16, 32, 5, 7, 3), (64, 64, 1, 1, 4), (48, 20, 11, 6, 1), (256, 128, 32, 32, 1)])
def test_k4_s2_p1_polyphase_winograd_f5x5_2x2_vs_fp64(Kc, Cf, nh, nw, B):
    """The 4x4 stride-2 pad-1 layers (Conv2d of netP/netD/netF, ConvTranspose2d of netP/netG; models/networks.py:235-243,
    404-432) by Winograd F(5x5,2x2) on the polyphase components: all three modes against fp64, read both as a Conv2d
    (fine = x, coarse = y) and as a ConvTranspose2d (coarse = x, fine = y), on grids that are and are not multiples of the
    5x5 tile, down to the 1x1 bottleneck."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Kc * 7 + nh)
    fine = torch.randn(B, Cf, 2 * nh, 2 * nw, generator=g).cuda()
    coarse = torch.randn(B, Kc, nh, nw, generator=g).cuda()
    w = (torch.randn(Kc, Cf, 4, 4, generator=g) * 0.1).cuda()
    fd, cd, wd = fine.double().cpu().requires_grad_(True), coarse.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    # as a Conv2d: y = conv(fine, w); as a ConvTranspose2d: y = convT(coarse, w)
    y64 = F.conv2d(fd, wd, None, 2, 1)
    dx64, dw64 = torch.autograd.grad(y64, (fd, wd), cd.detach())
    z64 = F.conv_transpose2d(cd, wd, None, 2, 1)
    dc64, dwt64 = torch.autograd.grad(z64, (cd, wd), fd.detach())
    dw = ops.conv4x4s2_winograd(ops.S2_WEIGHT_GRAD, fine, coarse, B, Kc, Cf, nh, nw)
    assert _rel(dw, dw64) <= 1e-4 and _rel(dw, dwt64) <= 1e-4          # the same tensor in both readings
    if Cf % 4 == 0:
        y = ops.conv4x4s2_winograd(ops.S2_FINE_TO_COARSE, fine, w, B, Kc, Cf, nh, nw)
        assert _rel(y, y64.detach()) <= 1e-4                           # Conv2d forward
        assert _rel(y, dc64) <= 1e-4                                   # == ConvTranspose2d backward-data of `fine`
    else:
        assert not ops.s2_winograd_supported(ops.S2_FINE_TO_COARSE, B, Kc, Cf, nh, nw)
    if Kc % 16 == 0:
        z = torch.full((B, Cf, 2 * nh, 2 * nw), float("nan"), device="cuda")      # every element must be overwritten
        ops.conv4x4s2_winograd(ops.S2_COARSE_TO_FINE, coarse, w, B, Kc, Cf, nh, nw, out=z)
        assert _rel(z, z64.detach()) <= 1e-4                           # ConvTranspose2d forward
        assert _rel(z, dx64) <= 1e-4                                   # == Conv2d backward-data of `coarse`
    else:
        assert not ops.s2_winograd_supported(ops.S2_COARSE_TO_FINE, B, Kc, Cf, nh, nw)


@pytest.mark.parametrize("kind,Ci,Co,H,W,k,st,pad,dil,B", [
    ("conv", 128, 128, 8, 8, 4, 2, 1, 1, 3), ("conv", 128, 256, 4, 6, 3, 1, 1, 1, 2), ("conv", 128, 256, 8, 8, 4, 2, 3, 2, 2), ("conv", 128, 128, 2, 2, 4, 2, 1, 1, 8),
    ("convT", 128, 64, 4, 4, 4, 2, 1, 1, 2), ("convT", 256, 128, 1, 1, 4, 2, 1, 1, 8), ("convT", 128, 128, 5, 3, 3, 1, 1, 1, 2), ("convT", 256, 32, 8, 8, 4, 2, 1, 1, 1)])
def test_smallmap_engine_vs_fp64(kind, Ci, Co, H, W, k, st, pad, dil, B):
    """ipsr_conv_smallmap: the inner levels of the U-Nets / netF (models/networks.py:220-259, 404-432, 510-515) — input gradient of
    a Conv2d, forward of a ConvTranspose2d and the weight gradient of either, with the weight tensor read in place as the GEMM
    operand: within 2e-5 of fp64 on 1x1 ... 8x8 grids, k3 / k4, stride 1 / 2, dilated."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Ci + 3 * H + k)
    tr = kind == "convT"
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), generator=g) * 0.1).cuda()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    y64 = F.conv_transpose2d(xd, wd, None, st, pad, 0, 1, dil) if tr else F.conv2d(xd, wd, None, st, pad, dil)
    dy = torch.randn(y64.shape, generator=g).cuda()
    dx64, dw64 = torch.autograd.grad(y64, (xd, wd), dy.double().cpu())
    Hy, Wy = y64.shape[2:]
    if tr:       # weight [Ci][Co]: R = Ci on the x grid (Ho x Wo), Cq = Co on the y grid (Hf x Wf)
        geo = (B, Ci, Co, H, W, Hy, Wy, k, st, pad, dil)
        y = ops.conv_smallmap(ops.SM_DATA, x, w, *geo)
        assert _rel(y, y64.detach()) <= 2e-5
        dw = ops.conv_smallmap(ops.SM_WRW, x, dy, *geo)
    else:        # weight [Co][Ci]: R = Co on the y grid, Cq = Ci on the x grid
        geo = (B, Co, Ci, Hy, Wy, H, W, k, st, pad, dil)
        dx = torch.full((B, Ci, H, W), float("nan"), device="cuda")
        ops.conv_smallmap(ops.SM_DATA, dy, w, *geo, out=dx)
        assert _rel(dx, dx64) <= 2e-5
        dw = ops.conv_smallmap(ops.SM_WRW, dy, x, *geo)
    assert tuple(dw.shape) == tuple(w.shape) and _rel(dw, dw64) <= 2e-5
    # the third operation: Conv2d forward / ConvTranspose2d backward-data (weight rows x im2col of the fine tensor)
    if tr:
        dxt = torch.full((B, Ci, H, W), float("nan"), device="cuda")
        ops.conv_smallmap(ops.SM_FWD, dy, w, *geo, out=dxt)
        assert _rel(dxt, dx64) <= 2e-5
    else:
        yc = ops.conv_smallmap(ops.SM_FWD, x, w, *geo)
        assert _rel(yc, y64.detach()) <= 2e-5


@pytest.mark.parametrize("kind,Ci,Co,H,W,B", [("conv", 3, 64, 32, 48, 2), ("conv", 6, 64, 20, 36, 3), ("convT", 128, 3, 24, 32, 2), ("convT", 32, 6, 9, 12, 1),
                                              ("conv", 3, 16, 5, 7, 2), ("convT", 64, 3, 64, 256, 1)])
def test_thin_3x3_layers_vs_fp64(kind, Ci, Co, H, W, B):
    """ipsr_conv3x3_thin(_wrw): VGG conv1_1 (3 -> 64, + bias + ReLU), netG's first Conv2d (6 -> 64) and last ConvTranspose2d
    (128 -> 3) — forward, input gradient and weight gradient on the vector ALUs, within 2e-5 of fp64; odd sizes, W % 4 == 0 where
    the many -> few kernel requires it."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Ci * 11 + H)
    tr = kind == "convT"
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn((Ci, Co, 3, 3) if tr else (Co, Ci, 3, 3), generator=g) * 0.1).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    xd, wd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    y64 = F.conv_transpose2d(xd, wd, None, 1, 1) if tr else F.conv2d(xd, wd, None, 1, 1)
    dy = torch.randn(y64.shape, generator=g).cuda()
    dx64, dw64 = torch.autograd.grad(y64, (xd, wd), dy.double().cpu())
    fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
    if ops.thin_supported(fop, Ci, H, W, Co):
        y = ops.conv3x3_thin(fop, x, w, (B, Ci, H, W), Co)
        assert _rel(y, y64.detach()) <= 2e-5
        if not tr:          # few -> many with the VGG epilogue
            yb = ops.conv3x3_thin(fop, x, w, (B, Ci, H, W), Co, bias=bias, relu=True)
            assert _rel(yb, torch.relu(y64.detach() + bias.double().cpu().view(1, -1, 1, 1))) <= 2e-5
    else:
        assert W % 4 != 0
    if ops.thin_supported(bop, Ci, H, W, Co):
        dx = ops.conv3x3_thin(bop, dy, w, (B, Ci, H, W), Co)
        assert _rel(dx, dx64) <= 2e-5
    else:
        assert W % 4 != 0
    if W % 4 == 0:
        dw = ops.conv3x3_thin_wrw(tr, x, dy)
        assert tuple(dw.shape) == tuple(w.shape) and _rel(dw, dw64) <= 2e-5


@pytest.mark.parametrize("kind,Ci,Co,H,W,B", [("conv", 3, 64, 32, 48, 2), ("conv", 6, 64, 20, 36, 3), ("convT", 128, 3, 24, 32, 2), ("convT", 64, 3, 64, 256, 1)])
@pytest.mark.parametrize("in_bf16,out_bf16", [(True, True), (False, True), (True, False)])
def test_thin_3x3_layers_on_bf16_tensors(kind, Ci, Co, H, W, B, in_bf16, out_bf16):
    """ipsr_conv3x3_thin_io / _wrw_io with bf16 tensors on either side (BASELINE config 5): the arithmetic is autocast's — operands
    rounded to bf16 (weights and an fp32 input on the way in), fp32 accumulation, ONE rounding on the way out.  Against an fp64
    convolution of the rounded operands: fp32 results within 2e-5, bf16 results within half a bf16 ulp of the result's scale (2^-8)
    plus that."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Ci * 13 + H)
    tr = kind == "convT"
    bf = torch.bfloat16
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn((Ci, Co, 3, 3) if tr else (Co, Ci, 3, 3), generator=g) * 0.1).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    xin = x.to(bf) if in_bf16 else x
    xd, wd = x.to(bf).double().cpu().requires_grad_(True), w.to(bf).double().cpu().requires_grad_(True)
    y64 = F.conv_transpose2d(xd, wd, None, 1, 1) if tr else F.conv2d(xd, wd, None, 1, 1)
    dy = torch.randn(y64.shape, generator=g).cuda()
    dyin = dy.to(bf) if in_bf16 else dy
    dx64, dw64 = torch.autograd.grad(y64, (xd, wd), dy.to(bf).double().cpu())
    tol = 2e-5 + (2.0 ** -8 if out_bf16 else 0.0)
    odt = bf if out_bf16 else torch.float32
    fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
    y = ops.conv3x3_thin(fop, xin, w, (B, Ci, H, W), Co, out_dtype=odt)
    assert y.dtype == odt and _rel(y.float(), y64.detach()) <= tol
    if not tr:
        yb = ops.conv3x3_thin(fop, xin, w, (B, Ci, H, W), Co, bias=bias, relu=True, out_dtype=odt)
        assert _rel(yb.float(), torch.relu(y64.detach() + bias.double().cpu().view(1, -1, 1, 1))) <= tol
    dx = ops.conv3x3_thin(bop, dyin, w, (B, Ci, H, W), Co, out_dtype=odt)
    assert dx.dtype == odt and _rel(dx.float(), dx64) <= tol
    # weight gradient: fp32 result; operands (x, dy) = (first flag, second flag) bf16 or fp32-rounded-inside
    dw = ops.conv3x3_thin_wrw(tr, xin, dy.to(bf) if out_bf16 else dy)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == tuple(w.shape)
    if in_bf16 or out_bf16:                       # with both operands fp32 the kernel is the fp32 one (no rounding): covered above
        assert _rel(dw, dw64) <= 2e-5


@pytest.mark.parametrize("kind,Ci,Co,H,W,B,k,st", [("conv", 6, 64, 20, 48, 3, 3, 1), ("conv", 3, 64, 32, 64, 2, 4, 2), ("convT", 128, 3, 24, 32, 2, 3, 1),
                                                   ("convT", 128, 3, 16, 32, 2, 4, 2), ("conv", 3, 16, 8, 16, 1, 3, 1), ("convT", 72, 6, 12, 16, 2, 4, 2),
                                                   ("conv", 3, 64, 256, 256, 2, 4, 2)])
@pytest.mark.parametrize("small_bf16", [False, True, None])
def test_thin_weight_gradient_on_the_matrix_cores(kind, Ci, Co, H, W, B, k, st, small_bf16):
    """ipsr_conv_thin_wrw_mfma: dW of the k3 s1 p1 / k4 s2 p1 layers with 3 or 6 channels on the narrow side, reduction over pixels on
    v_mfma_f32_32x32x16_bf16 (bf16 wide tensor) or v_mfma_f32_32x32x2_f32 (fp32 tensors) — against the fp64 autograd gradient of the module on
    the SAME operands (bf16-rounded in the bf16 cases): within 2e-5 of the
    gradient's scale (fp32 accumulation of up to 131072 products per entry), and the same bits on a second call (fixed summation order)."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Ci * 7 + H + k)
    tr = kind == "convT"
    bf = torch.bfloat16
    fp32 = small_bf16 is None                     # third case: both tensors fp32, multiplied in fp32 (v_mfma_f32_32x32x2_f32, BASELINE config 2)
    rnd = (lambda t: t) if fp32 else (lambda t: t.to(bf))
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), generator=g) * 0.1).cuda()
    xd, wd = rnd(x).double().cpu(), w.double().cpu().requires_grad_(True)
    y64 = F.conv_transpose2d(xd, wd, None, st, 1) if tr else F.conv2d(xd, wd, None, st, 1)
    dy = torch.randn(y64.shape, generator=g).cuda()
    (dw64,) = torch.autograd.grad(y64, (wd,), rnd(dy).double().cpu())
    assert ops.thin_wrw_mfma_supported(tr, B, Ci, H, W, Co, k, st)
    # bf16 cases: the wide tensor is bf16; the narrow one bf16 or fp32 (rounded inside)
    xin = x if fp32 else (x.to(bf) if (tr or small_bf16) else x)
    dyin = dy if fp32 else (dy.to(bf) if (not tr or small_bf16) else dy)
    dw = ops.conv_thin_wrw_mfma(tr, xin, dyin, k, st)
    assert dw.dtype == torch.float32 and tuple(dw.shape) == tuple(w.shape)
    assert _rel(dw, dw64) <= 2e-5
    assert torch.equal(dw, ops.conv_thin_wrw_mfma(tr, xin, dyin, k, st))


@pytest.mark.parametrize("kind,Ci,Co,H,W,B,k,st", [("conv", 3, 64, 20, 64, 2, 3, 1), ("conv", 6, 64, 12, 32, 3, 3, 1), ("conv", 3, 64, 32, 64, 2, 4, 2),
                                                   ("convT", 128, 3, 24, 32, 2, 3, 1), ("convT", 72, 6, 8, 32, 2, 4, 2), ("conv", 3, 24, 8, 32, 1, 3, 1),
                                                   ("conv", 3, 64, 256, 256, 2, 3, 1), ("conv", 3, 64, 256, 256, 1, 4, 2)])
@pytest.mark.parametrize("in_bf16,out_bf16", [(False, True), (True, True), (True, False)])
def test_thin_few_to_many_on_the_matrix_cores(kind, Ci, Co, H, W, B, k, st, in_bf16, out_bf16):
    """ipsr_conv_thin_f2m_mfma: Conv2d forward (+ bias + ReLU) and ConvTranspose2d input gradient with 3 / 6 channels on the side that is
    read, k3 s1 p1 and k4 s2 p1, on v_mfma_f32_32x32x16_bf16 — against fp64 on the same bf16-rounded operands: fp32 results within 2e-5
    of the result's scale, bf16 results within that plus half a bf16 ulp (2^-8)."""
    from deepinpainting_amd import ops
    g = torch.Generator().manual_seed(Ci * 5 + H + k)
    tr = kind == "convT"
    bf = torch.bfloat16
    x = torch.randn(B, Ci, H, W, generator=g).cuda()
    w = (torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), generator=g) * 0.1).cuda()
    bias = torch.randn(Ci if tr else Co, generator=g).cuda()
    xd, wd = x.to(bf).double().cpu().requires_grad_(True), w.to(bf).double().cpu()
    y64 = F.conv_transpose2d(xd, wd, None, st, 1) if tr else F.conv2d(xd, wd, None, st, 1)
    tol = 2e-5 + (2.0 ** -8 if out_bf16 else 0.0)
    odt = bf if out_bf16 else torch.float32
    if tr:
        dy = torch.randn(y64.shape, generator=g).cuda()
        (ref,) = torch.autograd.grad(y64, (xd,), dy.to(bf).double().cpu())
        op, inp = ops.CONVT_BWD_DATA, (dy.to(bf) if in_bf16 else dy)
    else:
        ref, op, inp = y64.detach(), ops.CONV_FWD, (x.to(bf) if in_bf16 else x)
    assert ops.thin_f2m_mfma_supported(op, B, Ci, H, W, Co, k, st)
    y = ops.conv_thin_f2m_mfma(op, inp, w, (B, Ci, H, W), Co, k, st, out_dtype=odt)
    assert y.dtype == odt and tuple(y.shape) == tuple(ref.shape) and _rel(y.float(), ref) <= tol
    yb = ops.conv_thin_f2m_mfma(op, inp, w, (B, Ci, H, W), Co, k, st, bias=bias, relu=True, out_dtype=odt)
    assert _rel(yb.float(), torch.relu(ref + bias.double().cpu().view(1, -1, 1, 1))) <= tol


# ---- the training step's OWN shapes (BASELINE config 2: batch 8, 256x256) ------------------------------------------------------
# (module, input H=W): the layers the step spends its time in, every Winograd family, all three passes.  References: the same
# module in fp64 on the GPU (torch's native convolution) AND MIOpen fp32 on the same tensors.
@pytest.mark.parametrize("B,C,H,W,K,pad", [(16, 512, 31, 31, 4, 1), (8, 512, 31, 31, 4, 1), (3, 70, 9, 11, 3, 1), (2, 64, 20, 47, 4, 1), (1, 64, 5, 5, 4, 0)])
def test_conv_with_one_output_channel(B, C, H, W, K, pad):
    """ipsr_conv_to_one (netD's last layer, nn.Conv2d(512, 1, 4, 1, 1) on 31x31): forward and weight gradient against fp64, within
    fp32 summation noise (5e-6 of the result's scale), and through the module path (dispatcher -> "one", bias by the bias kernel)."""
    from deepinpainting_amd import ops
    from deepinpainting_amd.models import hipconv
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + C)
    x = torch.randn(B, C, H, W, device="cuda", generator=g)
    w = torch.randn(1, C, K, K, device="cuda", generator=g) * 0.05
    y = ops.conv_to_one(x, w, pad)
    yd = F.conv2d(x.double(), w.double(), None, 1, pad)
    assert y.shape == yd.shape
    assert float((y.double() - yd).abs().max()) <= 5e-6 * float(yd.abs().max())
    dy = torch.randn(y.shape, device="cuda", generator=g)
    dw = ops.conv_to_one_wrw(x, dy, K, pad)
    dwd = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    assert float((dw.double() - dwd).abs().max()) <= 5e-6 * float(dwd.abs().max())
    if C >= 64:
        m = nn.Conv2d(C, 1, K, 1, pad).cuda()
        xr = x.clone().requires_grad_(True)
        hipconv._FORCE = "auto"
        try:
            assert hipconv.select(ops.CONV_FWD, B, C, H, W, 1, K, 1, pad, 1) == "one"
            ym = hipconv.conv_nobias(m, xr)
            gx, gw = torch.autograd.grad(ym, (xr, m.weight), dy)
        finally:
            hipconv._FORCE = None
        xq = x.clone().requires_grad_(True)
        yq = F.conv2d(xq, m.weight, None, 1, pad)
        qx, qw = torch.autograd.grad(yq, (xq, m.weight), dy)
        torch.testing.assert_close(ym, yq, rtol=1e-4, atol=1e-4 * float(yq.abs().max()))
        torch.testing.assert_close(gw, qw, rtol=1e-4, atol=1e-4 * float(qw.abs().max()))
        torch.testing.assert_close(gx, qx, rtol=1e-4, atol=1e-4 * float(qx.abs().max()))


STEP_LAYERS = [
    ("k3_512_32",      lambda: nn.Conv2d(512, 512, 3, 1, 1), 32),                      # VGG conv4_x, netG downconv_3: tiles 16 -> head/tail cut
    ("k3T_1024_256_32", lambda: nn.ConvTranspose2d(1024, 256, 3, 1, 1), 32),           # netG upconv_3: tiles 8 -> head/tail cut
    ("k3_256_64",      lambda: nn.Conv2d(256, 256, 3, 1, 1), 64),                      # VGG conv3_x
    ("k3_128_128",     lambda: nn.Conv2d(128, 128, 3, 1, 1), 128),                     # VGG conv2_2
    ("k4s2_256_512_32", lambda: nn.Conv2d(256, 512, 4, 2, 1), 32),                     # netP / netF down
    ("k4s2T_512_128_32", lambda: nn.ConvTranspose2d(512, 128, 4, 2, 1), 32),           # netP / netG up
    ("k4s2_64_128_128", lambda: nn.Conv2d(64, 128, 4, 2, 1), 128),                     # netD
    ("k4d2_512_32",    lambda: nn.Conv2d(512, 512, 4, 2, 3, dilation=2), 32),          # netG dilated down
    ("k4d2_128_128",   lambda: nn.Conv2d(128, 128, 4, 2, 3, dilation=2), 128),
    ("k4s1_256_512_32", lambda: nn.Conv2d(256, 512, 4, 1, 1), 32),                     # netD stride-1
    ("k3_64_256",      lambda: nn.Conv2d(64, 64, 3, 1, 1), 256),                       # VGG conv1_2
    ("k3_64_128_128",  lambda: nn.Conv2d(64, 128, 3, 1, 1), 128),                      # VGG conv2_1
    ("k4s2T_128_64_128", lambda: nn.ConvTranspose2d(128, 64, 4, 2, 1), 64),            # netP / netG outer up convolution
    ("k3T_256_64_128", lambda: nn.ConvTranspose2d(256, 64, 3, 1, 1), 128),             # netG upconv_1: 64 produced channels -> the GEMM's 64-row tile
    ("k4d2_64_256",    lambda: nn.Conv2d(64, 64, 4, 2, 3, dilation=2), 256),           # netG outermost dilated down convolution (64-row tile)
    ("k4s2T_64_64_128", lambda: nn.ConvTranspose2d(64, 64, 4, 2, 1), 128),             # netG outermost up convolution: row-writing output transform
    ("k4s1_512_1_31",  lambda: nn.Conv2d(512, 1, 4, 1, 1), 31),                        # netD's last layer ("one")
]


@pytest.mark.parametrize("name,make,H", STEP_LAYERS, ids=[t[0] for t in STEP_LAYERS])
def test_step_shapes_batch8_all_three_passes(name, make, H):
    """Forward, input gradient and weight gradient of the step's dominant layers at ITS batch (8) through the dispatcher's own
    choice ("auto"), within 1e-4 of the result's scale of an fp64 evaluation and of MIOpen fp32 on the same tensors."""
    from deepinpainting_amd.models import hipconv
    torch.manual_seed(len(name) * 7 + H)
    m = make().cuda()
    with torch.no_grad():
        m.weight.mul_(0.5)
    B = 8
    x = torch.randn(B, m.in_channels, H, H, device="cuda", requires_grad=True)
    hipconv._FORCE = "auto"
    try:
        y = hipconv.conv_nobias(m, x)
        dy = torch.randn_like(y)
        dx, dw = torch.autograd.grad(y, (x, m.weight), dy)
    finally:
        hipconv._FORCE = None
    tr = isinstance(m, nn.ConvTranspose2d)
    f = (lambda a, w: F.conv_transpose2d(a, w, None, m.stride, m.padding, 0, 1, m.dilation)) if tr else (lambda a, w: F.conv2d(a, w, None, m.stride, m.padding, m.dilation))
    xr, wr = x.detach().clone().requires_grad_(True), m.weight.detach().clone().requires_grad_(True)
    ym = f(xr, wr)                                                    # MIOpen fp32
    dxm, dwm = torch.autograd.grad(ym, (xr, wr), dy)
    xd, wd = x.detach().double().requires_grad_(True), m.weight.detach().double().requires_grad_(True)
    yd = f(xd, wd)                                                    # fp64
    dxd, dwd = torch.autograd.grad(yd, (xd, wd), dy.double())

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())
    errs = dict(y=rel(y, yd), dx=rel(dx, dxd), dw=rel(dw, dwd), y_mi=rel(y, ym), dx_mi=rel(dx, dxm), dw_mi=rel(dw, dwm),
                miopen_y=rel(ym, yd), miopen_dx=rel(dxm, dxd), miopen_dw=rel(dwm, dwd))
    print(name, {k: "%.1e" % v for k, v in errs.items()})
    for k in ("y", "dx", "dw"):
        assert errs[k] <= 1e-4, (name, k, errs)
        assert errs[k + "_mi"] <= 2e-4, (name, k, errs)              # two fp32 results, each within 1e-4 of the truth


def test_step_shapes_take_both_reduction_cuts():
    """The head/tail cut (GEMMs of the first 32 Winograd points uncut, the last 4 cut 2-4 ways) is what the 512-channel 32x32
    layers of the step run on, the uniform cut what the weight gradients run on: the shapes of the test above reach both
    (the round-2 tests reached neither at the step's batch)."""
    import ctypes
    from deepinpainting_amd import _lib
    L = _lib.lib()

    def split(rows, cols, red):
        out = (ctypes.c_int * 5)()
        _lib.check(L.ipsr_wino_gemm_split(rows, cols, red, ctypes.cast(out, ctypes.c_void_p)), "ipsr_wino_gemm_split")
        return list(out)
    # 512 -> 512 @32x32, batch 8: K = 512 rows, T = 8 * 8 * 8 = 512 tiles, reduction 512 channels
    ns, sps, xs, nt, spt = split(512, 512, 512)
    assert xs == 32 and ns == 1 and nt >= 2, "512@32x32 forward is expected on the head/tail cut, got %s" % ([ns, sps, xs, nt, spt],)
    # ConvTranspose2d 1024 -> 256 @32x32: 256 rows, 512 tiles, reduction 1024
    ns, sps, xs, nt, spt = split(256, 512, 1024)
    assert xs == 32 and nt >= 2
    # 256 @64x64 forward: 2 x 16 = 32 tiles of 128x128 per point -> uniform rule
    ns2, _, xs2, nt2, _ = split(256, 2048, 256)
    assert xs2 == 36 and ns2 == nt2
    # a long reduction over few output tiles (weight gradient of 128 -> 128 @128x128: 8192 tiles) must be cut uniformly
    ns3, _, xs3, nt3, _ = split(128, 128, 8192)
    assert xs3 == 36 and ns3 == nt3 and ns3 > 1


@pytest.mark.parametrize("engine", ["direct", "winograd", "auto"])
def test_module_path_forward_and_gradients(engine):
    """models/hipconv.py: Conv2d / ConvTranspose2d modules through the dispatcher with one engine forced — output, input
    gradient and weight gradient against fp64, and against the plain torch module on the same weights."""
    from deepinpainting_amd import ops
    from deepinpainting_amd.models import hipconv
    torch.manual_seed(3)
    cases = [(nn.Conv2d(128, 128, 4, 2, 3, dilation=2), 32, 32), (nn.Conv2d(64, 128, 3, 1, 1), 16, 16), (nn.ConvTranspose2d(128, 64, 3, 1, 1), 16, 16),
             (nn.Conv2d(32, 32, 4, 2, 3, dilation=2), 16, 16), (nn.ConvTranspose2d(32, 16, 4, 2, 1), 8, 8), (nn.Conv2d(16, 32, 4, 2, 1), 16, 16),
             (nn.Conv2d(64, 128, 4, 2, 1), 32, 32), (nn.ConvTranspose2d(128, 64, 4, 2, 1), 16, 16), (nn.Conv2d(128, 256, 4, 1, 1), 16, 16),
             (nn.Conv2d(256, 256, 4, 2, 1), 4, 4), (nn.ConvTranspose2d(256, 256, 4, 2, 1), 2, 2), (nn.Conv2d(256, 256, 3, 1, 1), 2, 2),
             (nn.ConvTranspose2d(512, 256, 3, 1, 1), 2, 2), (nn.Conv2d(256, 256, 4, 2, 3, dilation=2), 4, 4),
             (nn.Conv2d(3, 64, 3, 1, 1), 64, 64), (nn.ConvTranspose2d(128, 3, 3, 1, 1), 64, 64)]
    hipconv._FORCE = engine
    try:
        for m, H, W in cases:
            m = m.cuda()
            x = torch.randn(2, m.in_channels, H, W, device="cuda", requires_grad=True)
            y = hipconv.conv_nobias(m, x)
            dy = torch.randn_like(y)
            dx, dw = torch.autograd.grad(y, (x, m.weight), dy)
            kind = "convT" if isinstance(m, nn.ConvTranspose2d) else "conv"
            y64, dx64, dw64 = _ref64(kind, x.detach(), m.weight.detach(), dy, m.stride[0], m.padding[0], m.dilation[0])
            assert _rel(y, y64) <= 1e-4 and _rel(dx, dx64) <= 1e-4 and _rel(dw, dw64) <= 1e-4, (engine, m)
            tr = isinstance(m, nn.ConvTranspose2d)
            eng = hipconv.select(ops.CONVT_FWD if tr else ops.CONV_FWD, 2, m.in_channels, H, W, m.out_channels, m.kernel_size[0], m.stride[0],
                                 m.padding[0], m.dilation[0])
            if eng != "miopen":              # this repo's kernels are deterministic; MIOpen may switch solvers between calls
                with torch.no_grad():
                    y2 = hipconv.conv_nobias(m, x.detach())
                assert torch.equal(y2, y.detach()), (engine, m)
    finally:
        hipconv._FORCE = None


# ---- the split-bf16 arithmetic (BASELINE config 5's "bf16 MFMA for ... convs") --------------------------------------------------
def _f64(f, x, w, dy):
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = f(xd, wd)
    dx, dw = torch.autograd.grad(y, (xd, wd), dy.double())
    return y.detach(), dx, dw


def _relerr(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("math,io_bf16", [("bf16x6", False), ("bf16x3", False), ("bf16x3", True), ("fp32", True)])
def test_split_bf16_winograd_arithmetic_all_families(math, io_bf16):
    """Every Winograd family (3x3 stride 1: Conv2d / ConvTranspose2d; 4x4 dilated and stride 1; 4x4 stride 2 polyphase) in all three
    passes with the transformed operands split into bf16 planes and multiplied on the bf16 matrix cores — and with bf16
    activation tensors in / out (config 5) — against fp64 on the GPU (of the bf16-rounded operands where io is bf16):
        bf16x6 : the fp32 path's bound, 1e-4 of the result's scale        bf16x3 : 1e-3 (measured ~1.5e-4)
        bf16 outputs: + the rounding of the result itself, 2^-8."""
    from deepinpainting_amd import ops
    tol = {"fp32": 1e-4, "bf16x6": 1e-4, "bf16x3": 1e-3}[math]
    otol = tol + (2.0 ** -8 if io_bf16 else 0.0)
    act = torch.bfloat16 if io_bf16 else torch.float32
    g = torch.Generator(device="cuda").manual_seed(17)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, device="cuda", generator=g) * scale).to(act)
    # --- 3x3 stride 1
    for tr, Cin, H, W, Cout, B in ((False, 64, 16, 16, 96, 2), (False, 128, 32, 32, 160, 3), (True, 64, 12, 20, 48, 2), (False, 48, 9, 13, 80, 2)):
        x, dy = rnd(B, Cin, H, W), rnd(B, Cout, H, W)
        w = torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3), device="cuda", generator=g) * 0.1
        f = (lambda a, ww: F.conv_transpose2d(a, ww, None, 1, 1)) if tr else (lambda a, ww: F.conv2d(a, ww, None, 1, 1))
        y64, dx64, dw64 = _f64(f, x, w, dy)
        fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
        y = ops.conv3x3_winograd(fop, x, w, (B, Cin, H, W), Cout, math=math)
        dx = ops.conv3x3_winograd(bop, dy, w, (B, Cin, H, W), Cout, math=math)
        dw = ops.conv3x3_winograd_wrw(tr, x, dy, Cout, math=math)
        assert y.dtype == act and dx.dtype == act and dw.dtype == torch.float32
        assert _relerr(y, y64) <= otol and _relerr(dx, dx64) <= otol and _relerr(dw, dw64) <= tol, ("k3", tr, Cin, H, W, Cout, _relerr(y, y64), _relerr(dx, dx64), _relerr(dw, dw64))
    # --- 4x4 dilated stride 2 (geom 0) and 4x4 stride 1 pad 1 (geom 1)
    for geom, Cin, H, W, Cout, B in ((0, 128, 32, 32, 128, 2), (0, 32, 16, 16, 48, 2), (1, 64, 16, 16, 128, 2), (1, 32, 7, 10, 32, 3)):
        st_, pad, dil = (2, 3, 2) if geom == 0 else (1, 1, 1)
        Ho, Wo = (H // 2, W // 2) if geom == 0 else (H - 1, W - 1)
        x, dy = rnd(B, Cin, H, W), rnd(B, Cout, Ho, Wo)
        w = torch.randn(Cout, Cin, 4, 4, device="cuda", generator=g) * 0.1
        y64, dx64, dw64 = _f64(lambda a, ww: F.conv2d(a, ww, None, st_, pad, dil), x, w, dy)
        y = ops.conv4x4_dilated_winograd(0, x, w, (B, Cin, H, W), Cout, geom=geom, math=math)
        dx = ops.conv4x4_dilated_winograd(1, dy, w, (B, Cin, H, W), Cout, geom=geom, math=math)
        dw = ops.conv4x4_dilated_winograd(2, x, dy, (B, Cin, H, W), Cout, geom=geom, math=math)
        assert _relerr(y, y64) <= otol and _relerr(dx, dx64) <= otol and _relerr(dw, dw64) <= tol, ("k4", geom, Cin, H, W, Cout, _relerr(y, y64), _relerr(dx, dx64), _relerr(dw, dw64))
    # --- 4x4 stride 2 pad 1, read as a Conv2d (fine = x) — the ConvTranspose2d reading is the same three calls
    for Kc, Cf, nh, nw, B in ((128, 64, 16, 16, 2), (64, 32, 8, 8, 3), (256, 128, 32, 32, 1), (48, 24, 11, 6, 1)):
        fine, coarse = rnd(B, Cf, 2 * nh, 2 * nw), rnd(B, Kc, nh, nw)
        w = torch.randn(Kc, Cf, 4, 4, device="cuda", generator=g) * 0.1
        y64, dx64, dw64 = _f64(lambda a, ww: F.conv2d(a, ww, None, 2, 1), fine, w, coarse)
        y = ops.conv4x4s2_winograd(ops.S2_FINE_TO_COARSE, fine, w, B, Kc, Cf, nh, nw, math=math)
        dx = ops.conv4x4s2_winograd(ops.S2_COARSE_TO_FINE, coarse, w, B, Kc, Cf, nh, nw, math=math)
        dw = ops.conv4x4s2_winograd(ops.S2_WEIGHT_GRAD, fine, coarse, B, Kc, Cf, nh, nw, math=math)
        assert _relerr(y, y64) <= otol and _relerr(dx, dx64) <= otol and _relerr(dw, dw64) <= tol, ("s2", Kc, Cf, nh, nw, _relerr(y, y64), _relerr(dx, dx64), _relerr(dw, dw64))


def test_every_engine_call_of_a_training_step_checked_in_situ(tmp_path):
    """One full optimize_parameters() at BASELINE config 2's batch (8 x 256x256, dropout on, as bench.py runs it): EVERY
    convolution call that goes through a HIP engine — forward, input gradient and weight gradient, on the tensors the step really
    produces — is recomputed by MIOpen on the same operands and compared: within 1e-4 of the result's scale.  (A whole-net
    gradient comparison cannot discriminate: netG's backward amplifies fp32 forward noise to 0.7 % between two MIOpen-only runs,
    tests/test_gpu_model.py::test_all_four_nets_gradients_...; per call there is nothing to amplify.)  Also proves which engines
    the step runs on."""
    import contextlib
    import io
    import torch.nn.functional as F_
    from deepinpainting_amd.models import hipconv
    from deepinpainting_amd.options import Option
    from deepinpainting_amd.models.models import create_model
    opt = Option(gpu_ids=[0], batchSize=8, use_dropout=True, quiet=True, allow_random_vgg=True, checkpoints_dir=str(tmp_path))
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(opt)
    g = torch.Generator(device="cuda").manual_seed(21)
    img = torch.rand(8, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    ref = torch.rand(8, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    seen = {}

    def hook(kind, engine, geom, operands, result):
        if engine == "miopen":
            return
        transposed, k, stride, pad, dil, Cout = geom
        with torch.no_grad():
            if kind == "forward":
                x, w = operands
                want = F_.conv_transpose2d(x, w, None, stride, pad, 0, 1, dil) if transposed else F_.conv2d(x, w, None, stride, pad, dil)
            else:
                dy, x, w = operands
                which = [kind == "input_grad", kind == "weight_grad", False]
                out = torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [pad, pad], [dil, dil], transposed, [0, 0], 1, which)
                want = out[0] if kind == "input_grad" else out[1]
            scale = float(want.abs().max())
            err = float((result.float() - want).abs().max()) / max(scale, 1e-30)
        key = (kind, engine, tuple(operands[-2].shape) if kind != "forward" else tuple(operands[0].shape), tuple(w.shape), stride, pad, dil)
        seen[key] = max(seen.get(key, 0.0), err)

    for step in range(2):               # the second step runs on weights Adam has moved (real, non-initial statistics)
        hipconv._check_hook = hook if step == 1 else None
        try:
            m.set_input(img, mask, ref)
            m.set_ref_latent()
            m.set_gt_latent()
            m.optimize_parameters()
        finally:
            hipconv._check_hook = None
    torch.cuda.synchronize()
    engines = {}
    for (kind, engine, xs, ws_, st, pd, dl), err in sorted(seen.items(), key=lambda kv: -kv[1]):
        engines.setdefault((kind, engine), []).append(err)
    worst = sorted(seen.items(), key=lambda kv: -kv[1])[:12]
    print("engine calls checked: %d distinct (kind, engine, shape)" % len(seen))
    for (kind, engine), errs in sorted(engines.items()):
        print("  %-12s %-9s %3d shapes, worst %.1e" % (kind, engine, len(errs), max(errs)))
    for key, err in worst:
        print("  worst: %.2e %s" % (err, key))
    assert len(seen) >= 60
    for need in (("forward", "winograd"), ("input_grad", "winograd"), ("weight_grad", "winograd"), ("forward", "wino_s2"), ("input_grad", "wino_s2"),
                 ("weight_grad", "wino_s2"), ("forward", "wino_dil"), ("input_grad", "wino_dil"), ("weight_grad", "wino_dil"),
                 ("weight_grad", "smallmap"), ("forward", "smallmap"), ("input_grad", "smallmap"), ("input_grad", "direct")):
        assert need in engines, "the step did not run %s on the %s engine" % need
    bad = [(k, e) for k, e in seen.items() if not e <= 1e-4]
    assert not bad, bad


@pytest.mark.parametrize("tr,Cin,H,W,Cout,B", [(False, 32, 16, 16, 48, 2), (False, 128, 32, 32, 160, 3), (True, 64, 32, 64, 48, 2), (False, 16, 128, 128, 16, 1),
                                               (True, 256, 8, 32, 304, 2), (False, 64, 64, 64, 64, 2), (True, 144, 64, 16, 208, 1), (False, 512, 32, 32, 512, 16), (False, 64, 256, 256, 64, 1), (True, 32, 4, 256, 16, 2),
                                               (False, 256, 16, 16, 256, 2), (True, 512, 16, 16, 272, 3)])          # (the last two: reduction cut into four runs)
def test_direct_bf16_conv_all_passes(tr, Cin, H, W, Cout, B):
    """csrc/conv_bf16.hip (BASELINE config 5): the k3 s1 p1 layers as direct implicit GEMMs on v_mfma_f32_32x32x16_bf16 — forward, input
    gradient (bf16 and fp32 outputs) and weight gradient — against fp64 of the SAME bf16-rounded operands: what is left is the fp32
    accumulation (measured <= 2e-6) plus, for bf16 outputs, the rounding of the result, 2^-8.  Shapes: every supported width, channel
    counts that are not tile multiples, non-square maps, the step's largest layer at batch 16."""
    from deepinpainting_amd import ops
    g = torch.Generator(device="cuda").manual_seed(29)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(B, Cout, H, W, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3), device="cuda", generator=g) * 0.1
    f = (lambda a, ww: F.conv_transpose2d(a, ww, None, 1, 1)) if tr else (lambda a, ww: F.conv2d(a, ww, None, 1, 1))
    y64, dx64, _ = _f64(f, x, w.to(torch.bfloat16), dy)            # the kernel rounds the weights to bf16
    _, _, dw64 = _f64(f, x, w, dy)                                 # the weight gradient does not depend on the weights
    fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
    assert ops.conv3x3_bf16_supported(fop, B, Cin, H, W, Cout)
    y32 = ops.conv3x3_bf16(fop, x, w, (B, Cin, H, W), Cout, out_dtype=torch.float32)
    dx32 = ops.conv3x3_bf16(bop, dy, w, (B, Cin, H, W), Cout, out_dtype=torch.float32)
    y16 = ops.conv3x3_bf16(fop, x, w, (B, Cin, H, W), Cout)
    dx16 = ops.conv3x3_bf16(bop, dy, w, (B, Cin, H, W), Cout)
    assert y16.dtype == torch.bfloat16 and dx16.dtype == torch.bfloat16
    errs = dict(y=_relerr(y32, y64), dx=_relerr(dx32, dx64), y16=_relerr(y16, y64), dx16=_relerr(dx16, dx64))
    assert errs["y"] <= 1e-5 and errs["dx"] <= 1e-5, errs
    assert errs["y16"] <= 2.0 ** -8 and errs["dx16"] <= 2.0 ** -8, errs
    # frozen weights: the packed image is kept and the second call (no packing launch) gives the same bits
    with torch.no_grad():
        k1 = ops.conv3x3_bf16(fop, x, w, (B, Cin, H, W), Cout, keep_packed=True)
        k2 = ops.conv3x3_bf16(fop, x, w, (B, Cin, H, W), Cout, keep_packed=True)
    assert torch.equal(k1, y16) and torch.equal(k2, y16)
    if W == 256:                     # one image row per tile: forward forms only (VGG conv1_2); the weight gradient stops at 128
        assert not ops.conv3x3_bf16_wrw_supported(tr, B, Cin, H, W, Cout)
        return
    dw = ops.conv3x3_bf16_wrw(tr, x, dy, Cout)
    assert dw.dtype == torch.float32 and dw.shape == w.shape and _relerr(dw, dw64) <= 1e-5, _relerr(dw, dw64)
    # writes into a caller's buffer (a gradient bucket slice) and refuses what it cannot do
    sink = torch.full_like(w, float("nan"))
    assert ops.conv3x3_bf16_wrw(tr, x, dy, Cout, out=sink) is sink and torch.equal(sink, dw)
    assert not ops.conv3x3_bf16_supported(fop, B, Cin, H, 24, Cout) and not ops.conv3x3_bf16_supported(fop, B, Cin + 3, H, W, Cout)
    with pytest.raises(NotImplementedError):
        ops.conv3x3_bf16(fop, torch.zeros(1, 16, 12, 24, device="cuda", dtype=torch.bfloat16), torch.zeros((16, 16, 3, 3), device="cuda"), (1, 16, 12, 24), 16)


@pytest.mark.parametrize("Kc,Cf,nh,nw,B", [(128, 64, 64, 64, 2), (48, 32, 16, 16, 3), (256, 128, 16, 32, 2), (200, 16, 32, 32, 1), (64, 64, 4, 64, 2), (512, 128, 32, 32, 16),
                                           (256, 256, 16, 16, 2), (272, 512, 16, 16, 3)])                                # (the last two: split reductions)
def test_direct_bf16_stride2_family(Kc, Cf, nh, nw, B):
    """The k4 s2 p1 layers on the direct bf16 kernels: fine -> coarse (Conv2d forward / ConvTranspose2d input gradient: the input row
    parity is a sub-stage, the column parities two planes of the LDS image) and coarse -> fine (ConvTranspose2d forward / Conv2d input
    gradient: one workgroup per output row parity, the two column parities interleaved in the store) against fp64 of the bf16-rounded
    operands — fp32 and bf16 outputs, non-square grids, channel counts off the tile sizes."""
    from deepinpainting_amd import ops
    g = torch.Generator(device="cuda").manual_seed(31)
    fine = torch.randn(B, Cf, 2 * nh, 2 * nw, device="cuda", generator=g).to(torch.bfloat16)
    coarse = torch.randn(B, Kc, nh, nw, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(Kc, Cf, 4, 4, device="cuda", generator=g) * 0.1
    wd = w.to(torch.bfloat16).double()
    c64 = F.conv2d(fine.double(), wd, None, 2, 1)
    f64 = F.conv_transpose2d(coarse.double(), wd, None, 2, 1)
    if Cf % 16 == 0:
        c32 = ops.conv4x4s2_bf16(ops.S2_FINE_TO_COARSE, fine, w, B, Kc, Cf, nh, nw, out_dtype=torch.float32)
        c16 = ops.conv4x4s2_bf16(ops.S2_FINE_TO_COARSE, fine, w, B, Kc, Cf, nh, nw)
        assert c16.dtype == torch.bfloat16 and _relerr(c32, c64) <= 1e-5 and _relerr(c16, c64) <= 2.0 ** -8, (_relerr(c32, c64), _relerr(c16, c64))
    if Kc % 16 == 0:
        f32 = ops.conv4x4s2_bf16(ops.S2_COARSE_TO_FINE, coarse, w, B, Kc, Cf, nh, nw, out_dtype=torch.float32)
        f16 = ops.conv4x4s2_bf16(ops.S2_COARSE_TO_FINE, coarse, w, B, Kc, Cf, nh, nw)
        assert f16.dtype == torch.bfloat16 and _relerr(f32, f64) <= 1e-5 and _relerr(f16, f64) <= 2.0 ** -8, (_relerr(f32, f64), _relerr(f16, f64))
    # weight gradient (reduction over the coarse pixels; the fine windows are every other pixel of a row)
    if nw <= 64:
        dw64 = torch.ops.aten.convolution_backward(coarse.double(), fine.double(), wd, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        dw = ops.conv4x4s2_bf16_wrw(fine, coarse, B, Kc, Cf, nh, nw)
        assert dw.dtype == torch.float32 and dw.shape == w.shape and _relerr(dw, dw64) <= 1e-5, _relerr(dw, dw64)
        sink = torch.full_like(w, float("nan"))
        assert ops.conv4x4s2_bf16_wrw(fine, coarse, B, Kc, Cf, nh, nw, out=sink) is sink and torch.equal(sink, dw)
    assert not ops.conv4x4s2_bf16_supported(ops.S2_FINE_TO_COARSE, B, Kc, Cf, nh, 24) and not ops.conv4x4s2_bf16_supported(ops.S2_WEIGHT_GRAD, B, Kc, Cf, nh, nw)


def test_direct_bf16_conv_through_the_modules_under_autocast():
    """models/hipconv.py with bf16 activations: the k3 s1 p1 modules run forward, input gradient and weight gradient on the direct
    bf16 engine ("bf16d") from 32x32 maps up — asserted through the dispatcher's own hook — and match the fp32 module to bf16 accuracy."""
    from deepinpainting_amd import ops
    from deepinpainting_amd.models import hipconv
    torch.manual_seed(5)
    seen = []
    hipconv._check_hook = lambda kind, eng, geom, operands, result: seen.append((kind, eng))
    try:
        for m, H, W in ((nn.Conv2d(128, 256, 3, 1, 1), 64, 64), (nn.ConvTranspose2d(256, 128, 3, 1, 1), 32, 32), (nn.Conv2d(128, 256, 4, 2, 1), 128, 128),
                        (nn.ConvTranspose2d(256, 64, 4, 2, 1), 64, 64)):
            m = m.cuda()
            x = torch.randn(8, m.in_channels, H, W, device="cuda", requires_grad=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = hipconv.conv_nobias(m, x)
            assert y.dtype == torch.bfloat16
            dy = torch.randn_like(y, dtype=torch.float32)
            dx, dw = torch.autograd.grad(y.float(), (x, m.weight), dy)
            xr = x.detach().clone().requires_grad_(True)
            yr = F.conv_transpose2d(xr, m.weight, None, m.stride, m.padding) if isinstance(m, nn.ConvTranspose2d) else F.conv2d(xr, m.weight, None, m.stride, m.padding)
            dxr, dwr = torch.autograd.grad(yr, (xr, m.weight), dy)
            for a, b in ((y.float(), yr), (dx, dxr), (dw, dwr)):
                assert float((a - b).abs().max() / b.abs().max()) <= 2e-2          # bf16 operands and results: 2^-8 each
    finally:
        hipconv._check_hook = None
    assert seen.count(("forward", "bf16d")) == 4 and seen.count(("input_grad", "bf16d")) == 4 and seen.count(("weight_grad", "bf16d")) == 4, seen
    assert hipconv.select(ops.CONV_FWD, 16, 512, 16, 16, 512, 3, 1, 1, 1, True) == "bf16d"             # 16x16: the direct kernel with its reduction cut over workgroups
    assert hipconv.select(ops.CONV_FWD, 16, 512, 32, 32, 512, 3, 1, 1, 1, True) == "bf16d" and hipconv.select_wrw(False, 16, 512, 16, 16, 512, 3, 1, 1, 1, True) == "winograd"
    assert hipconv.select(ops.CONV_FWD, 16, 128, 128, 128, 128, 3, 1, 1, 1, True) == "bf16d"
    assert hipconv.select_wrw(False, 16, 64, 128, 128, 128, 3, 1, 1, 1, True) == "bf16d" and hipconv.select_wrw(False, 16, 64, 256, 256, 64, 3, 1, 1, 1, True) == "miopen"
    assert hipconv.select(ops.CONV_FWD, 16, 128, 128, 128, 128, 3, 1, 1, 1, False) == "winograd"       # fp32 activations: untouched


def test_dispatcher_rules_and_refusals():
    from deepinpainting_amd import ops
    from deepinpainting_amd.models import hipconv
    sel = hipconv.select
    assert sel(ops.CONV_FWD, 8, 512, 32, 32, 512, 3, 1, 1, 1) == "winograd"          # VGG conv4_x / netG level 32x32
    assert sel(ops.CONVT_BWD_DATA, 8, 1024, 32, 32, 256, 3, 1, 1, 1) == "winograd"
    assert sel(ops.CONV_FWD, 8, 64, 256, 256, 64, 3, 1, 1, 1) == "winograd"          # 64 -> 64 at 256x256 (VGG conv1_2): the GEMM's 64-row tile
    assert hipconv.select_wrw(False, 8, 64, 256, 256, 64, 3, 1, 1, 1) == "miopen"    # its weight gradient stays: the tile-major transforms lose
    assert sel(ops.CONV_FWD, 8, 3, 256, 256, 64, 3, 1, 1, 1) == "thin"               # VGG conv1_1: vector-ALU pass
    assert sel(ops.CONV_BWD_DATA, 8, 3, 256, 256, 64, 3, 1, 1, 1) == "thin" and sel(ops.CONVT_FWD, 8, 128, 256, 256, 3, 3, 1, 1, 1) == "thin"
    assert sel(ops.CONV_FWD, 8, 6, 256, 256, 64, 3, 1, 1, 1) == "miopen"             # 6 -> 64: MIOpen is faster
    assert sel(ops.CONV_FWD, 8, 512, 4, 4, 512, 3, 1, 1, 1) == "miopen"              # tiny maps
    assert sel(ops.CONV_BWD_DATA, 8, 512, 32, 32, 512, 4, 2, 3, 2) == "wino_dil"         # netG dilated down convolution
    assert sel(ops.CONV_FWD, 8, 128, 128, 128, 128, 4, 2, 3, 2) == "wino_dil"
    assert sel(ops.CONV_FWD, 8, 64, 256, 256, 64, 4, 2, 3, 2) == "wino_dil"          # outermost dilated level on the 64-row tile
    assert hipconv.select_wrw(False, 8, 64, 256, 256, 64, 4, 2, 3, 2) == "miopen"
    assert sel(ops.CONV_BWD_DATA, 8, 512, 16, 16, 512, 4, 2, 3, 2) == "direct"
    assert hipconv.select_wrw(False, 8, 256, 64, 64, 256, 4, 2, 3, 2) == "wino_dil"
    assert sel(ops.CONV_FWD, 8, 256, 32, 32, 512, 4, 1, 1, 1) == "wino_dil"              # netD's 4x4 stride-1 convolution
    assert sel(ops.CONV_BWD_DATA, 8, 256, 32, 32, 512, 4, 1, 1, 1) == "wino_dil"
    assert hipconv.select_wrw(False, 8, 256, 32, 32, 512, 4, 1, 1, 1) == "wino_dil"
    assert sel(ops.CONV_FWD, 8, 512, 31, 31, 1, 4, 1, 1, 1) == "one"                  # its one-channel head
    assert hipconv.select_wrw(False, 8, 512, 32, 32, 512, 3, 1, 1, 1) == "winograd"
    assert hipconv.select_wrw(False, 8, 128, 128, 128, 128, 3, 1, 1, 1) == "miopen"
    # 4x4 stride-2 pad-1: polyphase Winograd from 128 coarse / 64 fine channels up on coarse grids of 16..64
    assert sel(ops.CONV_FWD, 8, 256, 32, 32, 512, 4, 2, 1, 1) == "wino_s2"               # netP / netF down 256 -> 512 @32 -> 16
    assert sel(ops.CONV_BWD_DATA, 8, 64, 128, 128, 128, 4, 2, 1, 1) == "wino_s2"         # netD 64 -> 128 @128 -> 64
    assert sel(ops.CONVT_FWD, 8, 512, 32, 32, 128, 4, 2, 1, 1) == "wino_s2"              # netP up 512 -> 128 @32 -> 64
    assert sel(ops.CONVT_BWD_DATA, 8, 1024, 16, 16, 256, 4, 2, 1, 1) == "wino_s2"
    assert hipconv.select_wrw(True, 8, 256, 32, 32, 256, 4, 2, 1, 1) == "wino_s2"
    assert hipconv.select_wrw(False, 8, 128, 64, 64, 256, 4, 2, 1, 1) == "wino_s2"
    assert sel(ops.CONVT_FWD, 8, 64, 128, 128, 64, 4, 2, 1, 1) == "wino_s2"              # 64 channels at 128x128: only the pass whose output leaves as rows
    assert sel(ops.CONVT_BWD_DATA, 8, 64, 128, 128, 64, 4, 2, 1, 1) == "miopen"          # the other two are transform bound
    assert hipconv.select_wrw(True, 8, 64, 128, 128, 64, 4, 2, 1, 1) == "miopen"
    assert sel(ops.CONV_FWD, 8, 512, 16, 16, 512, 4, 2, 1, 1) == "miopen"                # 8x8 coarse grid: too few tiles
    assert sel(ops.CONV_BWD_DATA, 16, 512, 16, 16, 512, 4, 2, 1, 1) == "wino_s2" and sel(ops.CONV_BWD_DATA, 8, 512, 16, 16, 512, 4, 2, 1, 1) == "miopen"
    assert sel(ops.CONV_FWD, 8, 3, 256, 256, 64, 4, 2, 1, 1) == "miopen"                 # 3 input channels
    # innermost levels: the weight gradient of the 4x4 stride-2 layers as one GEMM that writes dW in place
    assert hipconv.select_wrw(False, 8, 512, 8, 8, 512, 4, 2, 1, 1) == "smallmap"        # netP down 512 -> 512 @8 -> 4
    assert hipconv.select_wrw(True, 8, 1024, 4, 4, 512, 4, 2, 1, 1) == "smallmap"        # netP up 1024 -> 512 @4 -> 8
    assert hipconv.select_wrw(False, 8, 512, 8, 8, 512, 4, 2, 3, 2) == "smallmap"        # netG dilated down @8 -> 4
    assert hipconv.select_wrw(True, 8, 512, 8, 8, 512, 4, 2, 1, 1) == "miopen"           # 512 positions: no gain measured
    assert hipconv.select_wrw(False, 8, 512, 4, 4, 512, 3, 1, 1, 1) == "miopen"          # 3x3: MIOpen ties or wins
    assert sel(ops.CONV_FWD, 8, 512, 4, 4, 512, 4, 2, 1, 1) == "smallmap"                # <= 32 positions: forward and input gradient too
    assert sel(ops.CONVT_BWD_DATA, 8, 512, 1, 1, 512, 4, 2, 1, 1) == "smallmap"
    assert sel(ops.CONVT_FWD, 8, 1024, 2, 2, 512, 3, 1, 1, 1) == "smallmap"
    assert sel(ops.CONV_BWD_DATA, 8, 512, 8, 8, 512, 4, 2, 1, 1) == "miopen"             # 128 positions: a real GEMM, MIOpen ties
    with pytest.raises(NotImplementedError):
        ops.conv2d(ops.CONV_FWD, torch.zeros(1, 3, 8, 8, device="cuda"), torch.zeros(4, 3, 3, 3, device="cuda"), (1, 3, 8, 8), 4, 3, 1, 1, 1)
    with pytest.raises(RuntimeError):
        ops.conv2d(ops.CONV_FWD, torch.zeros(1, 4, 8, 8, device="cuda"), torch.zeros(4, 8, 3, 3, device="cuda"), (1, 4, 8, 8), 4, 3, 1, 1, 1)


def test_vgg_and_unet_outputs_unchanged_by_the_engines():
    """The whole VGG16 feature pass and a netP (unet_256) forward/backward with the HIP engines — the direct implicit GEMM forced
    everywhere, then the dispatcher's own choices — against the same nets on MIOpen only (netG with the IPSR layer and netD / netF:
    tests/test_gpu_model.py::test_all_four_nets_gradients_auto_engines_vs_miopen_with_one_truncation):
    features within 1e-4 of their scale, parameter gradients within 1e-3 of their own scale plus 3e-4 of the largest
    gradient's (fp32 summation-order noise through 16 levels of convolution + InstanceNorm backward, which cancels the large
    components: the small gradients of the outer levels sit on that noise floor)."""
    import contextlib
    import io
    from deepinpainting_amd.models import hipconv, networks
    from deepinpainting_amd.models.vgg16 import Vgg16
    from deepinpainting_amd.options import Option
    torch.manual_seed(77)                    # the comparison below sits on a noise floor: same data every run
    vgg = Vgg16().cuda().eval()
    x = torch.rand(2, 3, 128, 128, device="cuda") * 2 - 1
    outs = {}
    for eng in ("miopen", "auto"):
        hipconv._FORCE = eng
        try:
            with torch.no_grad():
                outs[eng] = vgg(x)
        finally:
            hipconv._FORCE = None
    for a, b in zip(outs["auto"], outs["miopen"]):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    opt = Option(gpu_ids=[0], use_dropout=False)
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    with contextlib.redirect_stdout(io.StringIO()):
        netP = networks.define_G(3, 3, 64, 'unet_256', opt, mask, 'instance', False, 'normal', [0], 0.02)[0]
    img = torch.rand(2, 3, 256, 256, device="cuda") * 2 - 1
    grads = {}
    for eng in ("miopen", "direct", "auto"):
        hipconv._FORCE = eng
        try:
            netP.zero_grad()
            netP(img).square().mean().backward()
            grads[eng] = [p.grad.clone() for p in netP.parameters()]
        finally:
            hipconv._FORCE = None
    # (a conv bias in front of an InstanceNorm has an exactly-zero true gradient: what is left there is rounding noise of the
    # size of the largest gradients' last bits, hence the global term)
    gmax = max(float(b.abs().max()) for b in grads["miopen"])
    for eng in ("direct", "auto"):          # the one-launch implicit GEMM everywhere; the dispatcher's own per-shape choices
        for a, b in zip(grads[eng], grads["miopen"]):
            assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 3e-4 * gmax, eng
