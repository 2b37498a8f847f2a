#!/usr/bin/env python3
"""Run the hand-written convolution engines a few times at one of the step's shapes (for rocprofv3 / PMC passes).

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o conv -- python3 tools/profile_conv.py --shape vgg4
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops  # noqa: E402

SHAPES = {"vgg4": (512, 32, 512), "vgg3": (256, 64, 256), "vgg2": (128, 128, 128), "g32": (256, 32, 512), "up32": (1024, 32, 256),
          "vgg1": (64, 256, 64)}      # vgg1: the 64-row GEMM tile (wino_gemm_kernel<64>)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="vgg4", choices=sorted(SHAPES))
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bf16-direct", action="store_true", help="the direct bf16 kernels (csrc/conv_bf16.hip) on bf16 activations instead")
    a = ap.parse_args()
    Cin, H, Cout = SHAPES[a.shape]
    B = a.batch
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, Cin, H, H, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
    dy = torch.randn(B, Cout, H, H, device="cuda", generator=g)
    if a.bf16_direct:
        xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
        for _ in range(a.iters):
            ops.conv3x3_bf16(ops.CONV_FWD, xb, w, (B, Cin, H, H), Cout)
            ops.conv3x3_bf16(ops.CONV_BWD_DATA, dyb, w, (B, Cin, H, H), Cout)
            if ops.conv3x3_bf16_wrw_supported(False, B, Cin, H, H, Cout):
                ops.conv3x3_bf16_wrw(False, xb, dyb, Cout)
        torch.cuda.synchronize()
        print("%s: direct bf16 Conv2d(%d -> %d, k3 s1 p1) on [%d,%d,%d,%d], %d x (fwd, bwd-data, wrw)" % (a.shape, Cin, Cout, B, Cin, H, H, a.iters))
        return
    for _ in range(a.iters):
        ops.conv3x3_winograd(ops.CONV_FWD, x, w, (B, Cin, H, H), Cout)
        ops.conv3x3_winograd(ops.CONV_BWD_DATA, dy, w, (B, Cin, H, H), Cout)
        ops.conv3x3_winograd_wrw(False, x, dy, Cout)
        ops.conv2d(ops.CONV_FWD, x, w, (B, Cin, H, H), Cout, 3, 1, 1, 1)
    torch.cuda.synchronize()
    print("%s: Conv2d(%d -> %d, k3 s1 p1) on [%d,%d,%d,%d], %d x (winograd fwd, bwd-data, wrw; direct fwd)" % (a.shape, Cin, Cout, B, Cin, H, H, a.iters))


if __name__ == "__main__":
    main()
