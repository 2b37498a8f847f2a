#!/usr/bin/env python3
"""BASELINE config 5 (bf16 activations, batch 16): per layer shape of the training step, the three passes of the module path on
(a) MIOpen under bf16 autocast — including its NCHW<->NHWC transposes and the per-call weight casts — and (b) this repo's Winograd
engines reading / writing bf16 with split-bf16 operands on the bf16 MFMA (models/hipconv.py).  Decides `_bf16_wins`.

    python tools/bench_bf16.py [--batch 16]
"""
import argparse
import os
import statistics
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401  (private MIOpen db copy)
from deepinpainting_amd.models import hipconv

LAYERS = [
    ("k3 512->512 @32", lambda: nn.Conv2d(512, 512, 3, 1, 1), 32), ("k3 256->512 @32", lambda: nn.Conv2d(256, 512, 3, 1, 1), 32),
    ("k3 256->256 @64", lambda: nn.Conv2d(256, 256, 3, 1, 1), 64), ("k3 128->256 @64", lambda: nn.Conv2d(128, 256, 3, 1, 1), 64),
    ("k3 128->128 @128", lambda: nn.Conv2d(128, 128, 3, 1, 1), 128), ("k3 64->128 @128", lambda: nn.Conv2d(64, 128, 3, 1, 1), 128),
    ("k3 512->512 @16", lambda: nn.Conv2d(512, 512, 3, 1, 1), 16),
    ("k3T 1024->256 @32", lambda: nn.ConvTranspose2d(1024, 256, 3, 1, 1), 32), ("k3T 512->128 @64", lambda: nn.ConvTranspose2d(512, 128, 3, 1, 1), 64),
    ("k3T 256->64 @128", lambda: nn.ConvTranspose2d(256, 64, 3, 1, 1), 128), ("k3T 1024->512 @16", lambda: nn.ConvTranspose2d(1024, 512, 3, 1, 1), 16),
    ("k4s2 256->512 @32", lambda: nn.Conv2d(256, 512, 4, 2, 1), 32), ("k4s2 128->256 @64", lambda: nn.Conv2d(128, 256, 4, 2, 1), 64),
    ("k4s2 64->128 @128", lambda: nn.Conv2d(64, 128, 4, 2, 1), 128),
    ("k4s2T 512->128 @32", lambda: nn.ConvTranspose2d(512, 128, 4, 2, 1), 32), ("k4s2T 1024->256 @16", lambda: nn.ConvTranspose2d(1024, 256, 4, 2, 1), 16),
    ("k4s2T 256->64 @64", lambda: nn.ConvTranspose2d(256, 64, 4, 2, 1), 64), ("k4s2T 128->128 @64", lambda: nn.ConvTranspose2d(128, 128, 4, 2, 1), 64),
    ("k4d2 512->512 @32", lambda: nn.Conv2d(512, 512, 4, 2, 3, dilation=2), 32), ("k4d2 256->256 @64", lambda: nn.Conv2d(256, 256, 4, 2, 3, dilation=2), 64),
    ("k4d2 128->128 @128", lambda: nn.Conv2d(128, 128, 4, 2, 3, dilation=2), 128),
    ("k4s1 256->512 @32", lambda: nn.Conv2d(256, 512, 4, 1, 1), 32),
]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    B = a.batch
    print("%-20s | %28s | %28s | %s" % ("layer (batch %d)" % B, "MIOpen bf16 autocast  f / f+b", "split-bf16 Winograd  f / f+b", "ratio f+b"))
    for name, make, H in LAYERS:
        m = make().cuda()
        x = torch.randn(B, m.in_channels, H, H, device="cuda").bfloat16().requires_grad_(True)
        res = {}
        for eng in ("miopen", "auto"):
            hipconv._FORCE = eng

            def fwd():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return hipconv.conv_nobias(m, x)
            y = fwd()
            dy = torch.randn_like(y)

            def fwd_bwd():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    yy = hipconv.conv_nobias(m, x)
                torch.autograd.grad(yy, (x, m.weight), dy)
            with torch.no_grad():
                tf = timed(fwd)
            res[eng] = (tf, timed(fwd_bwd))
        hipconv._FORCE = None
        print("%-20s | %12.4f %12.4f ms | %12.4f %12.4f ms | %.2f" % (name, res["miopen"][0], res["miopen"][1], res["auto"][0], res["auto"][1],
                                                                    res["auto"][1] / res["miopen"][1]), flush=True)


if __name__ == "__main__":
    main()
