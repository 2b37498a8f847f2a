"""ctypes binding of libipsr_hip.so (C-ABI: include/ipsr_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C deepinpainting_amd/csrc`.
Loading fails LOUDLY: there is no fallback implementation of the layer.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libipsr_hip.so")
ABI_VERSION = 14

_lib = None

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/ipsr_hip.h exactly
SIGNATURES = {
    "ipsr_abi_version": (c_int, []),
    "ipsr_last_error": (ctypes.c_char_p, []),
    "ipsr_feat_mask_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ipsr_feat_mask": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ipsr_index_prep": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ipsr_patch_normalize": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ipsr_corr_argmax_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ipsr_corr_argmax": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "ipsr_bwd_index_ints": (c_size_t, [c_int, c_int]),
    "ipsr_forward_workspace_bytes": (c_size_t, [c_int] * 7),
    "ipsr_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ipsr_forward_masks": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    "ipsr_forward_bf16corr_workspace_bytes": (c_size_t, [c_int] * 7),
    "ipsr_forward_bf16corr": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ipsr_corr_argmax_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ipsr_corr_argmax_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ipsr_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int,
                              c_void_p, c_void_p]),
    "ipsr_backward_workspace_bytes": (c_size_t, [c_int] * 5),
    "ipsr_backward_patch": (c_int, [c_void_p, c_int, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "ipsr_cat_relu_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ipsr_cat_relu_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ipsr_bias_act": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "ipsr_bias_relu_pool2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ipsr_instnorm_act_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_float, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ipsr_instnorm_act_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                           c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ipsr_bias_act_skip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ipsr_bias_act_backward_skip": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "ipsr_instnorm_act_forward_slice": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_float, c_int, c_int, c_int, c_int,
                                                c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ipsr_instnorm_act_backward_slice": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                                 c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ipsr_bias_act_backward": (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "ipsr_conv2d_workspace_bytes": (c_size_t, [c_int] * 10),
    "ipsr_conv2d": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_winograd_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv3x3_winograd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_winograd_filter_floats": (c_size_t, [c_int, c_int, c_int]),
    "ipsr_conv3x3_winograd_ex": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_thin": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_long, ctypes.c_long, c_int, c_void_p]),
    "ipsr_conv3x3_thin_io": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_long, ctypes.c_long, c_int, c_int, c_void_p]),
    "ipsr_conv3x3_thin_wrw_io": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv_thin_f2m_mfma_supported": (c_int, [c_int] * 7),
    "ipsr_conv_thin_f2m_mfma": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_long, ctypes.c_long, c_int, c_int, c_void_p]),
    "ipsr_conv_thin_wrw_mfma_workspace_bytes": (c_size_t, [c_int] * 7),
    "ipsr_conv_thin_wrw_mfma": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_thin_wrw_workspace_bytes": (c_size_t, [c_int] * 5),
    "ipsr_conv_to_one_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv_to_one": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_thin_wrw": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv_smallmap_workspace_bytes": (c_size_t, [c_int] * 12),
    "ipsr_conv_smallmap": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4s2_winograd_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv4x4s2_winograd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4_winograd_workspace_bytes": (c_size_t, [c_int] * 7),
    "ipsr_conv4x4_winograd": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4_dilated_winograd_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv4x4_dilated_winograd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_winograd_wrw_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv3x3_winograd_wrw": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "innercos_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "innercos_loss": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    "innercos_loss_fused": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "innercos_loss_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                       c_void_p, c_void_p, c_void_p]),
    "ipsr_profile_enable": (c_int, [c_int]),
    "ipsr_profile_enable_mask": (c_int, [c_int, ctypes.c_uint]),
    "ipsr_profile_read": (c_int, [c_void_p, c_int]),
    "ipsr_profile_read_region": (c_int, [c_int, c_void_p, c_int]),
    "ipsr_profile_read_region_work": (c_int, [c_int, c_void_p, c_void_p, c_int]),
    "ipsr_profile_read_region_work2": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "ipsr_conv3x3_winograd_mp": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_winograd_wrw_mp": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4_winograd_mp": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4s2_winograd_mp": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_bf16_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv3x3_bf16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_bf16_packed": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4s2_bf16_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv4x4s2_bf16": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv4x4s2_bf16_wrw_workspace_bytes": (c_size_t, [c_int] * 5),
    "ipsr_conv4x4s2_bf16_wrw": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_conv3x3_bf16_wrw_workspace_bytes": (c_size_t, [c_int] * 6),
    "ipsr_conv3x3_bf16_wrw": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "ipsr_wino_gemm_split": (c_int, [c_int, c_int, c_int, c_void_p]),
    "ipsr_debug_force_wino_split": (c_int, [c_int, c_int, c_int]),
    "ipsr_debug_set_option": (c_int, [c_int, c_int]),
}


class IpsrLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises IpsrLibraryError if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IpsrLibraryError(
            "libipsr_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C deepinpainting_amd/csrc`.  The IPSR layer has no CPU/eager fallback." % LIB_PATH)
    h = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)            # AttributeError here = the .so does not match include/ipsr_hip.h
        fn.restype = res
        fn.argtypes = args
    if h.ipsr_abi_version() != ABI_VERSION:
        raise IpsrLibraryError("libipsr_hip.so ABI %d != expected %d — rebuild" % (h.ipsr_abi_version(), ABI_VERSION))
    _lib = h
    return _lib


def check(rc, what):
    """Map a C status to the exception the reference's Python would have raised."""
    if rc == 0:
        return
    msg = lib().ipsr_last_error().decode("utf-8", "replace")
    if rc == -2:
        raise NotImplementedError("%s: %s" % (what, msg))
    raise RuntimeError("%s failed (status %d): %s" % (what, rc, msg))
