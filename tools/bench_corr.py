#!/usr/bin/env python3
"""Time the correlation+arg-max kernel alone (HIP events through the C-ABI profiling hook)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import _lib, ops  # noqa: E402


def run(B, C, N, iters=30):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, C, N, device="cuda", generator=g).abs()
    ref = torch.relu(torch.randn(B, C, N, device="cuda", generator=g))
    xn, _ = ops.patch_normalize(x)
    L = _lib.lib()
    for _ in range(5):
        ops.corr_argmax(xn, ref)
    L.ipsr_profile_enable(iters)
    for _ in range(iters):
        ops.corr_argmax(xn, ref)
    buf = (ctypes.c_float * iters)()
    n = L.ipsr_profile_read(ctypes.cast(buf, ctypes.c_void_p), iters)
    L.ipsr_profile_enable(0)
    ms = sorted(buf[i] for i in range(n))
    med = ms[n // 2]
    fl = 2.0 * N * N * C * B
    print("B=%d C=%d N=%d: median %.1f us  min %.1f us  -> %.1f TFLOP/s (%.1f%% of 157.3)" %
          (B, C, N, med * 1e3, ms[0] * 1e3, fl / med / 1e9, 100 * fl / med / 1e9 / 157.3))


if __name__ == "__main__":
    run(8, 512, 1024)
    run(4, 512, 4096)
    run(16, 512, 1024)
