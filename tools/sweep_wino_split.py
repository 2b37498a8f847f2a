#!/usr/bin/env python3
"""Time the Winograd GEMM of one 3x3 layer under a forced reduction split (ipsr_debug_force_wino_split(nsplit, xi_split, nsplit_t),
honoured by wino_choose_split at every call) — the measurements behind the head/tail rule of csrc/winograd.hip.

    python tools/sweep_wino_split.py --shape 512,32,512 --splits 1,36,1 1,32,2 1,32,4 2,36,1
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", action="append", default=[], help="Cin,H,Cout (repeatable)")
    ap.add_argument("--splits", nargs="+", default=["auto", "1,36,1", "1,32,2", "1,32,4", "1,32,8", "2,36,1"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.lib()
    for shp in a.shape or ["512,32,512"]:
        Cin, H, Cout = map(int, shp.split(","))
        B = a.batch
        x = torch.randn(B, Cin, H, H, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        for sp in a.splits:
            _lib.check(lib.ipsr_debug_force_wino_split(*((0, 0, 0) if sp == "auto" else map(int, sp.split(",")))), "ipsr_debug_force_wino_split")
            for _ in range(3):
                ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, H), Cout)
            torch.cuda.synchronize()
            lib.ipsr_profile_enable_mask(a.iters, 0x8)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, H), Cout)
            e1.record()
            torch.cuda.synchronize()
            n = 256 * a.iters
            ms, work = (ctypes.c_float * n)(), (ctypes.c_double * n)()
            k = lib.ipsr_profile_read_region_work(3, ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(work, ctypes.c_void_p), n)
            lib.ipsr_profile_enable(0)
            g = statistics.median(ms[i] for i in range(k))
            lib.ipsr_debug_force_wino_split(0, 0, 0)
            print("%-14s split %-8s  GEMM %.4f ms (%.1f TF)   whole conv %.4f ms" % (shp, sp, g, work[0] / g / 1e9, e0.elapsed_time(e1) / a.iters), flush=True)


if __name__ == "__main__":
    main()
