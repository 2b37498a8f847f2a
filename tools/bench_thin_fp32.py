#!/usr/bin/env python3
"""Weight gradients of the thin layers in the fp32 training step (BASELINE config 2, batch 8): ipsr_conv_thin_wrw_mfma on fp32 tensors
(v_mfma_f32_32x32x2_f32) against aten.convolution_backward (MIOpen fp32, its transposes included).

    python tools/bench_thin_fp32.py > profiles/r04_thin_fp32.txt
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops  # noqa: E402
from bench_thin_bf16 import timeit  # noqa: E402


def main(B=8, iters=30):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    print("weight gradient, fp32 (batch %d)                    | MFMA thin ms | MIOpen fp32 ms | max err / max |ref| (fp64 of the fp32 operands)" % B)
    cases = [("netG first    conv  6 -> 64   k3 s1 @256", False, 6, 64, 256, 3, 1), ("netG last     convT 128 -> 3  k3 s1 @256", True, 128, 3, 256, 3, 1),
             ("netD / netP   conv  3 -> 64   k4 s2 @256", False, 3, 64, 256, 4, 2), ("netP last     convT 128 -> 3  k4 s2 @128", True, 128, 3, 128, 4, 2),
             ("netD (2B)     conv  3 -> 64   k4 s2 @256 x16", False, 3, 64, 256, 4, 2)]
    for name, tr, Cin, Cout, S, k, st in cases:
        b = 2 * B if "x16" in name else B
        x = torch.rand(b, Cin, S, S, device=dev, generator=g) * 2 - 1
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=dev, generator=g) * 0.1
        So = (S - 1) * st - 2 + k if tr else (S + 2 - k) // st + 1
        dy = torch.rand(b, Cout, So, So, device=dev, generator=g) * 2 - 1
        t_m = timeit(lambda: ops.conv_thin_wrw_mfma(tr, x, dy, k, st), iters)
        t_mi = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [1, 1], [1, 1], tr, [0, 0], 1, [False, True, False])[1], iters)
        xd, wd = x[:2].double().cpu(), w.double().cpu().requires_grad_(True)
        y64 = F.conv_transpose2d(xd, wd, None, st, 1) if tr else F.conv2d(xd, wd, None, st, 1)
        (ref,) = torch.autograd.grad(y64, (wd,), dy[:2].double().cpu())
        got = ops.conv_thin_wrw_mfma(tr, x[:2].contiguous(), dy[:2].contiguous(), k, st).double().cpu()
        print("%-51s | %12.4f | %14.4f | %.2e" % (name, t_m, t_mi, float((got - ref).abs().max() / ref.abs().max())), flush=True)


if __name__ == "__main__":
    main()
