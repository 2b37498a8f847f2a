#!/usr/bin/env python3
"""Time the correlation+arg-max kernel alone (HIP events through the C-ABI profiling hook)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import _lib, ops  # noqa: E402


def run(B, C, N, iters=30, zeros=False):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, C, N, device="cuda", generator=g).abs()
    ref = torch.relu(torch.randn(B, C, N, device="cuda", generator=g))
    xn, _ = ops.patch_normalize(x)
    if zeros:       # DVFS probe: all-zero operands draw less power, so the chip holds a higher clock
        xn.zero_(); ref.zero_()
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
    if hasattr(L, "ipsr_debug_read_probe"):      # diagnostic build (-DIPSR_CLOCK_PROBE): clock held inside the kernel
        import statistics
        nwg = min(8192, B * (N // 128) * max(1, min(N // 128, -(-512 // (B * (N // 128))))))
        pb = (ctypes.c_ulonglong * (2 * nwg))()
        torch.cuda.synchronize()
        L.ipsr_debug_read_probe(pb, nwg)
        clk = [pb[2 * i] / pb[2 * i + 1] * 100.0 for i in range(nwg) if pb[2 * i + 1] > 0]
        cyc = [pb[2 * i] for i in range(nwg) if pb[2 * i + 1] > 0]
        print("   in-kernel clock: median %.0f MHz (min %.0f max %.0f), K-loop cycles per workgroup median %.0f" %
              (statistics.median(clk), min(clk), max(clk), statistics.median(cyc)))
    print(("zeros " if zeros else "") + "B=%d C=%d N=%d: median %.1f us  min %.1f us  -> %.1f TFLOP/s (%.1f%% of 157.3)" %
          (B, C, N, med * 1e3, ms[0] * 1e3, fl / med / 1e9, 100 * fl / med / 1e9 / 157.3))


if __name__ == "__main__":
    run(8, 512, 1024)
    run(4, 512, 4096)
    run(16, 512, 1024)
    run(16, 512, 1024, zeros=True)
    run(4, 512, 4096, zeros=True)
