#!/usr/bin/env python3
"""Micro-timings of single C-ABI entry points (torch events, median of 30)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    B, C, N = 8, 512, 1024
    x = torch.randn(B, C, N, device="cuda").abs()
    y = torch.empty_like(x)
    print("patch_normalize (no xT): %.1f us" % timeit(lambda: ops.patch_normalize(x)))
    print("torch copy 16MB->16MB:   %.1f us" % timeit(lambda: y.copy_(x)))
    print("torch x*2 :              %.1f us" % timeit(lambda: torch.mul(x, 2.0, out=y)))
    z = torch.empty(B, N, C, device="cuda")
    print("torch transpose copy:    %.1f us" % timeit(lambda: z.copy_(x.transpose(1, 2))))


if __name__ == "__main__":
    main()
