"""The CPU oracle (oracle/ipsr_oracle.c) against the golden fixtures captured from the reference itself
(oracle/gen_golden.py).  This is what pins the oracle; HIP-vs-oracle parity is in test_gpu_parity.py.

Tolerances: index/byte work bit-exact; fp32 within 1e-4 (the north-star tolerance), in practice ~1e-6.
"""
import glob
import os

import numpy as np
import pytest

from oracle import ipsr_oracle as orc
import golden_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAYER_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "layer_*.npz")))
WELL_CONDITIONED = [c for c in LAYER_CASES if "signed" not in c]
ATOL = 1e-4


def load(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "x" not in d:  # the big case: inputs are regenerated from seeds
        d["x"], d["ref"] = golden_cases.big_case_inputs()
        d["grad_out"] = golden_cases.big_case_grad_out(name)
        d["ic_target"] = golden_cases.big_case_ic_target()
    return d


def test_fixture_inventory():
    assert len(LAYER_CASES) >= 9
    assert "layer_c512_8x8_cfg1" in LAYER_CASES and "layer_c512_32x32_cfg2" in LAYER_CASES


def test_masks_bit_exact():
    d = np.load(os.path.join(GOLDEN, "masks.npz"))
    tags = sorted({k.rsplit("__", 1)[0] for k in d.files if k.endswith("__mask")})
    assert len(tags) == 20
    for tag in tags:
        thr = float(tag.split("__thr")[1])
        feat = orc.feat_mask(d[tag + "__mask"], 3, thr)
        np.testing.assert_array_equal(feat, d[tag + "__feat"], err_msg=tag)
        ip = orc.index_prep(feat, 1, 1, 1)
        np.testing.assert_array_equal(ip.flag, d[tag + "__flag"])
        np.testing.assert_array_equal(ip.nonmask_point_idx, d[tag + "__nonmask"])
        np.testing.assert_array_equal(ip.mask_point_idx, d[tag + "__mask_point_idx"])
        np.testing.assert_array_equal(ip.flatten_offsets, d[tag + "__flatten_offsets"])


def test_reference_known_answer_M252():
    # util/NonparametricShift.py:17 comment "[252, 512, 1, 1]" for fineSize=256, overlap=4 (models/IPSR.py:40-41)
    m = np.zeros((256, 256), np.uint8)
    m[68:188, 68:188] = 1
    ip = orc.index_prep(orc.feat_mask(m, 3, 5 / 16.0))
    assert ip.mask_point_idx.shape[0] == 252
    m[:] = 0
    m[64:192, 64:192] = 1
    assert orc.index_prep(orc.feat_mask(m, 3, 5 / 16.0)).mask_point_idx.shape[0] == 256


@pytest.mark.parametrize("name", LAYER_CASES)
def test_mask_side_of_layer_cases(name):
    d = load(name)
    feat = orc.feat_mask(d["mask_img"], 3, float(d["threshold"]))
    np.testing.assert_array_equal(feat, d["feat_mask"])
    ip = orc.index_prep(feat)
    np.testing.assert_array_equal(ip.flag, d["flag"])
    np.testing.assert_array_equal(ip.mask_point_idx, d["mask_point_idx"])
    np.testing.assert_array_equal(ip.flatten_offsets, d["flatten_offsets"])


@pytest.mark.parametrize("name", WELL_CONDITIONED)
def test_forward_backward_vs_reference(name):
    d = load(name)
    B, C, h, w = d["x"].shape
    N = h * w
    f = orc.forward(d["x"], d["ref"], d["mask_point_idx"])
    # arg-max indices: exact (the fixtures have clear margins except the deliberate ties -> lowest index)
    np.testing.assert_array_equal(f.ind.astype(np.int64), d["ind"])
    np.testing.assert_allclose(f.vmax, d["vmax"], rtol=0, atol=ATOL)
    if "S" in d:
        xn, _ = orc.patch_normalize(d["x"].reshape(B, C, N))
        _, _, S = orc.corr_argmax(xn, d["ref"].reshape(B, C, N), want_S=True)
        np.testing.assert_allclose(S, d["S"], rtol=0, atol=ATOL)
    if "attn_rows" in d:
        np.testing.assert_allclose(f.attn_rows, d["attn_rows"], rtol=0, atol=ATOL)
    out = f.out if "out_channels" not in d else f.out[:, d["out_channels"]]
    err = np.abs(out - d["out"]).max()
    assert err <= ATOL, err
    # backward: the sparse trunc(kbar) has exactly as many entries as the reference's LongTensor has non-zeros
    M = d["mask_point_idx"].shape[0]
    for b in range(B):
        offA = f.bwd_index[b, :N + 1]
        offB = f.bwd_index[b, 2 * N + 1:3 * N + 2]
        assert offA[0] == 0 and offB[0] == 0 and (np.diff(offA) >= 0).all() and (np.diff(offB) >= 0).all()
        assert offA[N] == N - M and offA[N] + offB[N] == d["trunc_kbar_nnz"][b]
    gin = orc.backward(d["grad_out"], d["mask_point_idx"], f.attn_rows, f.bwd_index, float(d["triple_w"]))
    gin = gin if "grad_in_channels" not in d else gin[:, d["grad_in_channels"]]
    err = np.abs(gin - d["grad_in"]).max()
    assert err <= ATOL, err


def test_ill_conditioned_case_relative():
    # signed features: a/(a+vmax) is ill-conditioned -> indices exact, values by a relative metric
    d = load("layer_c16_8x8_signed")
    f = orc.forward(d["x"], d["ref"], d["mask_point_idx"])
    np.testing.assert_array_equal(f.ind.astype(np.int64), d["ind"])
    scale = np.abs(d["attn_rows"]).max()
    assert np.abs(f.attn_rows - d["attn_rows"]).max() <= 1e-3 * max(scale, 1.0)
    assert np.abs(f.out - d["out"]).max() <= 1e-3 * max(np.abs(d["out"]).max(), 1.0)
    gin = orc.backward(d["grad_out"], d["mask_point_idx"], f.attn_rows, f.bwd_index, float(d["triple_w"]))
    # trunc() of an ill-conditioned weight may flip an integer; require agreement on all but a few entries
    bad = np.abs(gin - d["grad_in"]) > 1e-3
    assert bad.mean() < 0.02


@pytest.mark.parametrize("name", LAYER_CASES)
def test_innercos_vs_reference(name):
    d = load(name)
    loss = orc.innercos_loss(d["x"], d["feat_mask"], d["ic_target"], float(d["strength"]))
    np.testing.assert_allclose(loss, d["ic_loss"], rtol=1e-5)
    if "ic_grad" in d:
        g = orc.innercos_loss_backward(d["x"], d["feat_mask"], d["ic_target"], float(d["strength"]))
        np.testing.assert_allclose(g, d["ic_grad"], rtol=1e-5, atol=1e-9)


def test_innercos2_narrow_512():
    d = np.load(os.path.join(GOLDEN, "innercos2_c1024_8x8.npz"))
    np.testing.assert_array_equal(orc.feat_mask(d["mask_img"], 3, float(d["threshold"])), d["feat_mask"])
    loss = orc.innercos_loss(d["x"], d["feat_mask"], d["ic_target"], float(d["strength"]))
    np.testing.assert_allclose(loss, d["ic_loss"], rtol=1e-5)
    g = orc.innercos_loss_backward(d["x"], d["feat_mask"], d["ic_target"], float(d["strength"]))
    np.testing.assert_allclose(g, d["ic_grad"], rtol=1e-5, atol=1e-9)
    assert not g[:, 512:].any()


def test_sps_helper():
    d = np.load(os.path.join(GOLDEN, "masks.npz"))
    sp_x = np.repeat(np.arange(5), 7)
    sp_y = np.tile(np.arange(7), 5)
    np.testing.assert_array_equal(sp_x, d["sps_5x7__sp_x"])
    np.testing.assert_array_equal(sp_y, d["sps_5x7__sp_y"])


# ---- shift_sz > 1 (BASELINE config 4's 3x3 patches) ----------------------------------------------------------------
PATCH_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "patch_layer_*.npz")))


def dense_trunc_kbar(f, mpi, b):
    """trunc(kbar) [N'(k), N'(q)] of sample b from the oracle's outputs (one-hot non-masked columns + truncated rows)."""
    Np = f.ind.shape[1]
    W = np.zeros((Np, Np), np.float32)
    masked = np.zeros(Np, bool)
    masked[mpi] = True
    q = np.nonzero(~masked)[0]
    W[f.ind[b, q], q] = 1.0
    for l, ql in enumerate(mpi):
        W[:, ql] = np.trunc(f.attn_rows[b, l])
    return W


def test_patch_fixture_inventory():
    assert len(PATCH_CASES) >= 3


@pytest.mark.parametrize("name", PATCH_CASES)
def test_patch_forward_vs_reference(name):
    """The reference's own forward for shift_sz > 1 (it computes the output and only then raises, IPSRFunction.py:134;
    gen_golden.py swallows that one store) against the oracle = unfold -> p=1 algorithm -> fold."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    p = int(d["patch"])
    ip = orc.index_prep(d["feat_mask"], patch=p)
    np.testing.assert_array_equal(ip.flag, d["flag"])
    np.testing.assert_array_equal(ip.mask_point_idx, d["mask_point_idx"])
    f = orc.forward(d["x"], d["ref"], d["mask_point_idx"], patch=p)
    np.testing.assert_array_equal(f.ind.astype(np.int64), d["ind"])
    np.testing.assert_allclose(f.vmax, d["vmax"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(f.attn_rows, d["attn_rows"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(f.out, d["out"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("name", PATCH_CASES)
def test_patch_unfold_fold_vs_torch(name):
    import torch
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    p = int(d["patch"])
    B, C, h, w = d["x"].shape
    xu = orc.unfold(d["x"], p)
    t = torch.nn.functional.unfold(torch.from_numpy(d["x"]), p)
    np.testing.assert_array_equal(xu, t.numpy())
    np.testing.assert_allclose(orc.fold(xu, C, h, w, p), torch.nn.functional.fold(t, (h, w), p).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", PATCH_CASES)
def test_patch_backward_extension_is_the_adjoint(name):
    """shift_sz > 1 backward (extension, no reference counterpart): grad_in = g + triple_w * d<out, g>/dx with kbar held
    constant and truncated — checked against torch autograd of fold(unfold(x) @ trunc(kbar))."""
    import torch
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    p = int(d["patch"])
    B, C, h, w = d["x"].shape
    mpi = d["mask_point_idx"]
    f = orc.forward(d["x"], d["ref"], mpi, patch=p)
    g = np.random.RandomState(5).standard_normal(d["x"].shape).astype(np.float32)
    tw = 0.75
    gin = orc.backward_patch(g, len(mpi), f.bwd_index, tw, p)
    for b in range(B):
        W = torch.from_numpy(dense_trunc_kbar(f, mpi, b)).double()
        x = torch.from_numpy(d["x"][b:b + 1]).double().requires_grad_(True)
        ou = torch.nn.functional.unfold(x, p)[0] @ W
        out = torch.nn.functional.fold(ou[None], (h, w), p)
        (out * torch.from_numpy(g[b:b + 1]).double()).sum().backward()
        want = g[b] + tw * x.grad[0].numpy()
        np.testing.assert_allclose(gin[b], want, rtol=1e-5, atol=1e-5)
