"""The C-ABI library loads and exports every symbol include/ipsr_hip.h declares (no GPU needed:
nothing here launches a kernel)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ipsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(?:int|size_t|const char\s*\*)\s+((?:ipsr|innercos)_\w+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from deepinpainting_amd import _lib
    return _lib


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("ipsr_forward", "ipsr_backward", "ipsr_feat_mask", "ipsr_index_prep", "ipsr_patch_normalize",
                 "ipsr_corr_argmax", "innercos_loss", "innercos_loss_backward", "ipsr_last_error"):
        assert must in names
    assert len(names) >= 15


def test_library_exports_every_declared_symbol(built):
    h = ctypes.CDLL(built.LIB_PATH)
    for name in declared_functions():
        assert hasattr(h, name), "libipsr_hip.so lacks %s" % name


def test_binding_table_matches_header(built):
    assert sorted(built.SIGNATURES) == declared_functions()
    L = built.lib()
    assert L.ipsr_abi_version() == built.ABI_VERSION


def test_host_side_queries_and_argument_errors(built):
    L = built.lib()
    assert L.ipsr_bwd_index_ints(1024, 256) == 2 * 1025 + 1024 + 2 * (256 * 257 // 2)
    assert L.ipsr_forward_workspace_bytes(8, 512, 32, 32, 256, 1, 1) > 8 * 512 * 1024 * 4 * 2
    # shift_sz = 3: the patch-major windows xT [8,900,4608], the un-folded result [8,4608,900] and the 1x1 correlation R [8,1024,1024]
    assert L.ipsr_forward_workspace_bytes(8, 512, 32, 32, 256, 3, 1) > (2 * 8 * 4608 * 900 + 8 * 1024 * 1024) * 4
    assert L.ipsr_forward_workspace_bytes(8, 512, 32, 32, 256, 3, 2) == 0          # stride != 1: unsupported
    assert L.ipsr_backward_workspace_bytes(8, 512, 32, 32, 1) == 0
    assert L.ipsr_backward_workspace_bytes(8, 512, 32, 32, 3) >= 2 * 8 * 4608 * 900 * 4
    assert L.ipsr_feat_mask_workspace_bytes(256, 256, 3) >= 2 * 128 * 128 * 4
    # argument validation happens before any HIP call, so it is testable without a GPU
    assert L.ipsr_forward(None, None, None, 0, 1, 1, 1, 1, 1, 1, None, None, None, None, None, None, 0, None) == -1
    assert b"null pointer" in L.ipsr_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = L.ipsr_forward(p, p, None, 0, 1, 1, 8, 8, 3, 2, p, p, p, None, None, p, 1 << 20, None)
    assert rc == -2 and b"stride=1" in L.ipsr_last_error()
    with pytest.raises(NotImplementedError):
        built.check(rc, "ipsr_forward")


def test_oracle_library_exports_cpu_twins():
    from oracle import ipsr_oracle as orc
    o = orc.lib()
    for name in ("ipsr_feat_mask", "ipsr_index_prep", "ipsr_patch_normalize", "ipsr_corr_argmax", "ipsr_forward",
                 "ipsr_backward", "ipsr_backward_patch", "ipsr_unfold", "ipsr_fold", "innercos_loss", "innercos_loss_backward"):
        assert hasattr(o, name + "_cpu")


def test_header_is_plain_c(tmp_path):
    """include/ipsr_hip.h is the C-ABI: it must compile as C99 (and as C++) on its own, with no HIP or torch headers."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "ipsr_hip.h"\nint (*take_address)(void) = ipsr_abi_version;\nint main(void) { return take_address == 0; }\n')
    for cc, flags in (("gcc", ["-std=c99", "-pedantic"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip("%s not available" % cc)
        subprocess.check_call([cc, "-Wall", "-Werror", "-I", os.path.join(root, "include")] + flags + ["-c", str(src), "-o", str(tmp_path / (cc + ".o"))])


def test_conv_dispatcher_rules_on_the_host():
    """models/hipconv.py picks an engine per (operation, shape) from measured rules plus workspace probes of the library — pure host
    logic (no kernel is launched): the step's main shapes land on the engines DESIGN.md §5.5 names, and shapes an engine cannot
    express are never routed to it."""
    from deepinpainting_amd import ops
    from deepinpainting_amd.models import hipconv
    sel, wrw = hipconv.select, hipconv.select_wrw
    assert sel(ops.CONV_FWD, 8, 512, 32, 32, 512, 3, 1, 1, 1) == "winograd"              # VGG conv4_x, netG level 32x32
    assert sel(ops.CONV_FWD, 8, 64, 256, 256, 64, 3, 1, 1, 1) == "winograd"              # 64 -> 64 at 256x256: the GEMM's 64-row tile
    assert wrw(False, 8, 64, 256, 256, 64, 3, 1, 1, 1) == "miopen"                       # its weight gradient: transform bound
    assert sel(ops.CONV_FWD, 16, 512, 31, 31, 1, 4, 1, 1, 1) == "one" and wrw(False, 16, 512, 31, 31, 1, 4, 1, 1, 1) == "one"   # netD's last layer
    assert sel(ops.CONV_BWD_DATA, 16, 512, 31, 31, 1, 4, 1, 1, 1) == "miopen"
    assert sel(ops.CONV_BWD_DATA, 8, 512, 32, 32, 512, 4, 2, 3, 2) == "wino_dil"         # netG dilated down convolution
    assert sel(ops.CONV_FWD, 8, 256, 32, 32, 512, 4, 1, 1, 1) == "wino_dil"              # netD's 4x4 stride-1 layer
    assert sel(ops.CONVT_FWD, 8, 512, 32, 32, 128, 4, 2, 1, 1) == "wino_s2"              # netP up 512 -> 128
    assert sel(ops.CONV_FWD, 8, 512, 4, 4, 512, 4, 2, 1, 1) == "smallmap"                # innermost level, 32 positions
    assert sel(ops.CONV_FWD, 8, 3, 256, 256, 64, 4, 2, 1, 1) == "miopen"                 # 3 input channels
    assert sel(ops.CONV_FWD, 8, 250, 32, 32, 512, 3, 1, 1, 1) == "miopen"                # reduction not a multiple of 16
    assert wrw(False, 8, 512, 32, 32, 512, 3, 1, 1, 1) == "winograd" and wrw(True, 8, 512, 32, 32, 128, 4, 2, 1, 1) == "wino_s2"
    assert wrw(False, 8, 512, 8, 8, 512, 4, 2, 1, 1) == "smallmap" and wrw(False, 8, 128, 128, 128, 128, 3, 1, 1, 1) == "miopen"
    # probes agree with the entry points' own argument checks
    assert ops.s2_winograd_supported(ops.S2_FINE_TO_COARSE, 8, 128, 64, 64, 64) and not ops.s2_winograd_supported(ops.S2_FINE_TO_COARSE, 8, 128, 3, 64, 64)
    assert ops.smallmap_supported(ops.SM_WRW, 8, 512, 512, 4, 4, 8, 8, 4, 2, 1, 1) and not ops.smallmap_supported(ops.SM_WRW, 8, 512, 500, 4, 4, 8, 8, 4, 2, 1, 1)
    assert not ops.smallmap_supported(ops.SM_DATA, 8, 512, 512, 4, 4, 11, 11, 4, 2, 1, 1)         # 11x11 does not map to a 4x4 output
    # bf16 activations (BASELINE config 5, batch 16): the direct kernels from 32x32 up, split-bf16 Winograd on 16x16 maps with >= 512
    # channels, the 3-channel ends on the stream kernels, MIOpen for what nothing here expresses
    assert sel(ops.CONV_FWD, 16, 512, 32, 32, 512, 3, 1, 1, 1, True) == "bf16d" and wrw(False, 16, 512, 32, 32, 512, 3, 1, 1, 1, True) == "bf16d"
    assert sel(ops.CONV_FWD, 16, 64, 256, 256, 64, 3, 1, 1, 1, True) == "bf16d"           # VGG conv1_2: the 64-row kernel, 512-pixel tile
    assert sel(ops.CONV_FWD, 16, 512, 16, 16, 512, 3, 1, 1, 1, True) == "bf16d" and wrw(False, 16, 512, 16, 16, 512, 3, 1, 1, 1, True) == "winograd"   # 16x16: split reduction
    assert sel(ops.CONV_FWD, 16, 256, 32, 32, 512, 4, 2, 1, 1, True) == "bf16d" and sel(ops.CONVT_FWD, 16, 1024, 16, 16, 256, 4, 2, 1, 1, True) == "bf16d"
    assert sel(ops.CONVT_BWD_DATA, 16, 1024, 16, 16, 256, 4, 2, 1, 1, True) == "miopen"                                                                    # 1024 produced: no room
    assert sel(ops.CONVT_FWD, 16, 512, 32, 32, 128, 4, 2, 1, 1, True) == "bf16d" and wrw(True, 16, 512, 32, 32, 128, 4, 2, 1, 1, True) == "bf16d"
    assert sel(ops.CONV_FWD, 16, 3, 256, 256, 64, 3, 1, 1, 1, True) == "thin"              # VGG conv1_1
    assert sel(ops.CONVT_BWD_DATA, 16, 128, 256, 256, 3, 3, 1, 1, 1, True) == "thin"       # the last ConvTranspose2d's input gradient (3 -> 128)
    assert sel(ops.CONVT_FWD, 16, 128, 256, 256, 3, 3, 1, 1, 1, True) == "bf16d"           # its forward: the direct kernel beats the stream kernel
    assert sel(ops.CONV_FWD, 16, 6, 256, 256, 64, 3, 1, 1, 1, True) == "miopen"            # 6 -> 64: MIOpen is ahead under bf16
    assert sel(ops.CONV_FWD, 16, 3, 256, 256, 64, 4, 2, 1, 1, True) == "thin_f2m"          # the first Conv2d of netP / netD
    assert sel(ops.CONV_FWD, 16, 64, 256, 256, 64, 4, 2, 3, 2, True) == "miopen"           # the dilated family on large maps
    assert wrw(False, 16, 3, 256, 256, 64, 4, 2, 1, 1, True) == "thin_mfma" and wrw(True, 16, 128, 256, 256, 3, 3, 1, 1, 1, True) == "thin_mfma"   # thin weight gradients
    assert wrw(False, 16, 6, 256, 256, 64, 3, 1, 1, 1, True) == "thin_mfma" and wrw(False, 8, 3, 256, 256, 64, 4, 2, 1, 1) == "thin_mfma"           # fp32 too (fp32 MFMA)
    assert wrw(True, 8, 128, 256, 256, 3, 3, 1, 1, 1) == "thin_mfma" and wrw(False, 8, 3, 32, 32, 64, 4, 2, 1, 1) == "miopen"                       # (small maps: MIOpen)
    assert ops.conv3x3_bf16_supported(ops.CONV_FWD, 2, 32, 16, 16, 48)                     # <= 64 produced channels on a map too small for 512-pixel tiles
    # the selection is memoised per shape and per forced mode; the A/B switches are read from the environment once (reload_env re-reads)
    was = hipconv._FORCE
    try:
        hipconv._FORCE = "miopen"
        assert sel(ops.CONV_FWD, 8, 512, 32, 32, 512, 3, 1, 1, 1) == "miopen" and sel(ops.CONV_FWD, 16, 512, 32, 32, 512, 3, 1, 1, 1, True) == "miopen"
        hipconv._FORCE = None
        assert sel(ops.CONV_FWD, 8, 512, 32, 32, 512, 3, 1, 1, 1) == "winograd"
        os.environ["IPSR_NO_THIN"] = "1"
        assert sel(ops.CONV_FWD, 16, 3, 256, 256, 64, 3, 1, 1, 1, True) == "thin"          # cached: the environment is not re-read per call
        hipconv.reload_env()
        assert sel(ops.CONV_FWD, 16, 3, 256, 256, 64, 3, 1, 1, 1, True) == "miopen"
    finally:
        hipconv._FORCE = was
        os.environ.pop("IPSR_NO_THIN", None)
        hipconv.reload_env()
