#!/usr/bin/env python3
"""Run the IPSR layer forward+backward a few times at a BASELINE.json config size (for rocprofv3).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o layer -- python3 tools/profile_layer.py --cfg 2
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops  # noqa: E402

CFG = {1: (2, 512, 8, 64, 16, 48), 2: (8, 512, 32, 256, 64, 192), 4: (4, 512, 64, 512, 128, 384)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--patch", type=int, default=1, help="shift_sz (BASELINE config 4 is --cfg 4 --patch 3)")
    ap.add_argument("--signed", action="store_true", help="signed features (what the conv stack really feeds the "
                    "layer): attention weights leave [0,1] and many survive the backward's truncation")
    a = ap.parse_args()
    B, C, h, size, lo, hi = CFG[a.cfg]
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(B, C, h, h, device="cuda", generator=g)
    if not a.signed:
        x = x.abs()
    ref = torch.relu(torch.randn(B, C, h, h, device="cuda", generator=g))
    grad = torch.randn(B, C, h, h, device="cuda", generator=g)
    m = torch.zeros(size, size, dtype=torch.uint8, device="cuda")
    m[lo:hi, lo:hi] = 1
    feat = ops.feat_mask(m, 3, 5 / 16.0)
    P = a.patch
    flag, mpi, cnt = ops.index_prep(feat, P, 1, 1)
    M = int(cnt.item())
    mpi = mpi[:M].contiguous()
    for _ in range(3):
        f = ops.forward(x, ref, mpi, patch=P)
        ops.backward(grad, f.bwd_index, 1.0, M, patch=P)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(a.iters):
        e[0].record()
        f = ops.forward(x, ref, mpi, patch=P)
        e[1].record()
        ops.backward(grad, f.bwd_index, 1.0, M, patch=P)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
    Np = (h - P + 1) ** 2
    flops = 2.0 * B * Np * Np * C * P * P
    print("cfg%d%s patch=%d B=%d C=%d %dx%d N'=%d M=%d: forward %.3f ms  backward %.3f ms (mean of %d); correlation %.1f GFLOP "
          "-> forward as a whole runs at %.1f TFLOP/s" %
          (a.cfg, " signed" if a.signed else "", P, B, C, h, h, Np, M, tf / a.iters, tb / a.iters, a.iters, flops / 1e9,
           flops / (tf / a.iters) / 1e9))


if __name__ == "__main__":
    main()
