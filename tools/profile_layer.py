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
    flag, mpi, cnt = ops.index_prep(feat, 1, 1, 1)
    M = int(cnt.item())
    mpi = mpi[:M].contiguous()
    for _ in range(3):
        f = ops.forward(x, ref, mpi)
        ops.backward(grad, f.bwd_index, 1.0, M)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(a.iters):
        e[0].record()
        f = ops.forward(x, ref, mpi)
        e[1].record()
        ops.backward(grad, f.bwd_index, 1.0, M)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
    print("cfg%d%s B=%d C=%d %dx%d M=%d: forward %.3f ms  backward %.3f ms (mean of %d)" %
          (a.cfg, " signed" if a.signed else "", B, C, h, h, M, tf / a.iters, tb / a.iters, a.iters))


if __name__ == "__main__":
    main()
