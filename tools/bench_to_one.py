#!/usr/bin/env python3
"""netD's last layer, nn.Conv2d(512, 1, 4, 1, 1) on 31x31: ipsr_conv_to_one against MIOpen, forward and weight gradient.

    python tools/bench_to_one.py
"""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401,E402
from deepinpainting_amd import ops  # noqa: E402


def t_ms(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)


for B in (8, 16):
    x = torch.randn(B, 512, 31, 31, device="cuda")
    w = torch.randn(1, 512, 4, 4, device="cuda") * 0.05
    dy = torch.randn(B, 1, 30, 30, device="cuda")
    print("B=%2d forward  one %.4f ms   miopen %.4f ms" % (B, t_ms(lambda: ops.conv_to_one(x, w, 1)), t_ms(lambda: F.conv2d(x, w, None, 1, 1))))
    print("B=%2d weight   one %.4f ms   miopen %.4f ms" % (B, t_ms(lambda: ops.conv_to_one_wrw(x, dy, 4, 1)), t_ms(
        lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))))
