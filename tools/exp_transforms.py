#!/usr/bin/env python3
"""A/B of transform-kernel variants (ipsr_debug_set_option) on the training step's 3x3 shapes: whole-convolution time by HIP
events, the GEMM's own time from the library's measurement hook (region 3) — the difference is what the transforms cost.

    python tools/exp_transforms.py --opt 1=0 --opt 1=1
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import _lib, ops  # noqa: E402

SHAPES = [(128, 128, 128), (64, 128, 128), (256, 64, 256), (128, 64, 256), (512, 64, 128), (512, 32, 512), (256, 32, 512), (1024, 32, 256), (512, 16, 512)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", action="append", default=[], help="key=value[,key=value] — one configuration per --opt")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--math", nargs="+", default=["fp32"], help="arithmetic(s) of the GEMM: fp32 bf16x3 bf16x6")
    ap.add_argument("--check", action="store_true", help="also report the error against an fp64 convolution (first two samples)")
    a = ap.parse_args()
    lib = _lib.lib()
    B = a.batch
    for Cin, H, Cout in SHAPES:
        x = torch.randn(B, Cin, H, H, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        for cfg, math in [(c, m_) for c in (a.opt or ["0=0"]) for m_ in a.math]:
            kv = [tuple(map(int, e.split("="))) for e in cfg.split(",")]
            for k, v in kv:
                _lib.check(lib.ipsr_debug_set_option(k, v), "ipsr_debug_set_option")
            for _ in range(3):
                y = ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, H), Cout, math=math)
            torch.cuda.synchronize()
            if a.check:
                y64 = torch.nn.functional.conv2d(x[:2].double(), w.double(), None, 1, 1)
                err = float((y[:2].double() - y64).abs().max() / y64.abs().max())
            else:
                err = float("nan")
            lib.ipsr_profile_enable_mask(a.iters, 0x8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, H), Cout, math=math)
            e1.record()
            torch.cuda.synchronize()
            n = 256 * a.iters
            ms, work = (ctypes.c_float * n)(), (ctypes.c_double * n)()
            k = lib.ipsr_profile_read_region_work(3, ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(work, ctypes.c_void_p), n)
            lib.ipsr_profile_enable(0)
            g = statistics.median(ms[i] for i in range(k))
            tot = e0.elapsed_time(e1) / a.iters
            for k_, _ in kv:
                lib.ipsr_debug_set_option(k_, 0)
            cfg = "%s %s" % (cfg, math)
            print("%4d->%4d @%3d  opt %-14s conv %.4f ms  GEMM %.4f (%.0f TF fp32-equivalent)  rest %.4f  err %.1e" %
                  (Cin, Cout, H, cfg, tot, g, work[0] / g / 1e9, tot - g, err), flush=True)


if __name__ == "__main__":
    main()
