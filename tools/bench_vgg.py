#!/usr/bin/env python3
"""Time one VGG16-features forward (B=8, 256x256, fp32) layer by layer on MIOpen."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa  (MIOpen db)
from deepinpainting_amd.models.vgg16 import Vgg16

v = Vgg16().cuda().eval()
x = torch.rand(8, 3, 256, 256, device="cuda") * 2 - 1
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    print("whole VGG forward: %.3f ms" % t(lambda: v(x)))
    h = x
    for s in (v.slice1, v.slice2, v.slice3, v.slice4):
        for name, m in s.named_children():
            inp = h
            ms = t(lambda: m(inp))
            h = m(inp)
            if isinstance(m, torch.nn.Conv2d):
                fl = 2 * m.in_channels * 9 * m.out_channels * h.shape[2] * h.shape[3] * 8
                print("  conv %3d->%3d @%3d: %.3f ms  %.1f TFLOP/s(direct-equivalent)" % (m.in_channels, m.out_channels, h.shape[2], ms, fl / ms / 1e9))
            else:
                print("  %s: %.3f ms" % (m.__class__.__name__, ms))
