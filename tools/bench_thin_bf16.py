#!/usr/bin/env python3
"""The 3x3 layers with a 3- or 6-channel side on bf16 activations (BASELINE config 5, batch 16): the vector-ALU stream kernels
(csrc/thin_conv.hip, ipsr_conv3x3_thin_io) against MIOpen under autocast (NHWC transposes and weight casts included) and, where it
applies, the direct bf16 MFMA kernel.  Prints ms per call (HIP events over `--iters` back-to-back calls) and the error against an
fp64 convolution of the bf16-rounded operands.

    python tools/bench_thin_bf16.py > profiles/r04_thin_bf16.txt
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    B, S = a.batch, 256
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    cases = [  # name, transposed, Cin, Cout, input dtype, passes
        ("VGG conv1_1   conv  3 -> 64 (+bias, ReLU)", False, 3, 64, torch.float32, ("fwd",)),
        ("netG first    conv  6 -> 64", False, 6, 64, torch.float32, ("fwd",)),
        ("netG last     convT 128 -> 3", True, 128, 3, torch.bfloat16, ("fwd", "bwd")),
    ]
    print("layer (batch %d, 256x256)                     | pass | thin bf16 ms | MIOpen autocast ms | direct MFMA ms | max err / max |ref|" % B)
    for name, tr, Cin, Cout, xdt, passes in cases:
        w = (torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3), device=dev, generator=g) * 0.1)
        bias = torch.randn(Cout, device=dev, generator=g) * 0.1
        x = (torch.rand(B, Cin, S, S, device=dev, generator=g) * 2 - 1).to(xdt)
        dy = (torch.rand(B, Cout, S, S, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        wr = w.to(torch.bfloat16).double()
        for ps in passes:
            fused = name.startswith("VGG")
            if ps == "fwd":
                op = ops.CONVT_FWD if tr else ops.CONV_FWD
                inp = x

                def thin():
                    return ops.conv3x3_thin(op, inp, w, (B, Cin, S, S), Cout, bias=bias if fused else None, relu=fused, out_dtype=torch.bfloat16)

                def miopen():
                    with torch.autocast("cuda", dtype=torch.bfloat16):
                        y = F.conv_transpose2d(inp, w, None, 1, 1) if tr else F.conv2d(inp, w, None, 1, 1)
                    if fused:
                        y = ops.bias_act_(y, bias, "relu")
                    return y
                xr = inp.to(torch.bfloat16).double()[:2]
                ref = F.conv_transpose2d(xr, wr, None, 1, 1) if tr else F.conv2d(xr, wr, None, 1, 1)
                if fused:
                    ref = torch.relu(ref + bias.double()[None, :, None, None])
            else:
                op = ops.CONVT_BWD_DATA if tr else ops.CONV_BWD_DATA
                inp = dy

                def thin():
                    return ops.conv3x3_thin(op, inp, w, (B, Cin, S, S), Cout, out_dtype=torch.bfloat16)
                wb = w.to(torch.bfloat16)
                xb = x.to(torch.bfloat16)

                def miopen():
                    return torch.ops.aten.convolution_backward(inp, xb, wb, None, [1, 1], [1, 1], [1, 1], tr, [0, 0], 1, [True, False, False])[0]
                dr = inp.double()[:2]
                ref = F.conv2d(dr, wr, None, 1, 1) if tr else F.conv_transpose2d(dr, wr, None, 1, 1)
            t_thin = timeit(thin, a.iters)
            t_mi = timeit(miopen, a.iters)
            t_dir = None
            if not fused and ops.conv3x3_bf16_supported(op, B, Cin, S, S, Cout):
                xin = inp.to(torch.bfloat16)
                t_dir = timeit(lambda: ops.conv3x3_bf16(op, xin, w, (B, Cin, S, S), Cout), a.iters)
            got = thin()[:2].double()
            err = float((got - ref).abs().max() / ref.abs().max())
            print("%-45s | %-4s | %12.4f | %18.4f | %14s | %.2e" % (name, ps, t_thin, t_mi, "%.4f" % t_dir if t_dir else "-", err), flush=True)


def wrw_table(B, iters):
    """Weight gradients of the thin layers: the MFMA kernel (ipsr_conv_thin_wrw_mfma) against aten.convolution_backward on bf16 operands
    (MIOpen, incl. its transposes; + the fp32 cast of dW)."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    print("\nweight gradient (batch %d)                          | MFMA thin ms | MIOpen bf16 ms | vector-ALU thin ms | max err / max |ref|" % B)
    cases = [("netG first    conv  6 -> 64   k3 s1 @256", False, 6, 64, 256, 3, 1), ("netG last     convT 128 -> 3  k3 s1 @256", True, 128, 3, 256, 3, 1),
             ("netD / netP   conv  3 -> 64   k4 s2 @256", False, 3, 64, 256, 4, 2), ("netP last     convT 128 -> 3  k4 s2 @128", True, 128, 3, 128, 4, 2)]
    for name, tr, Cin, Cout, S, k, st in cases:
        x = (torch.rand(B, Cin, S, S, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=dev, generator=g) * 0.1
        So = (S - 1) * st - 2 + k if tr else (S + 2 - k) // st + 1
        dy = (torch.rand(B, Cout, So, So, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        wb = w.to(torch.bfloat16)

        def mfma():
            return ops.conv_thin_wrw_mfma(tr, x, dy, k, st)

        def miopen():
            return torch.ops.aten.convolution_backward(dy, x, wb, None, [st, st], [1, 1], [1, 1], tr, [0, 0], 1, [False, True, False])[1].float()
        t_m = timeit(mfma, iters)
        t_mi = timeit(miopen, iters)
        t_v = timeit(lambda: ops.conv3x3_thin_wrw(tr, x, dy), iters) if k == 3 else None
        xd, wd = x[:2].double().cpu(), w.double().cpu().requires_grad_(True)
        y64 = F.conv_transpose2d(xd, wd, None, st, 1) if tr else F.conv2d(xd, wd, None, st, 1)
        (ref,) = torch.autograd.grad(y64, (wd,), dy[:2].double().cpu())
        got = ops.conv_thin_wrw_mfma(tr, x[:2].contiguous(), dy[:2].contiguous(), k, st).double().cpu()
        print("%-48s | %12.4f | %14.4f | %18s | %.2e" % (name, t_m, t_mi, "%.4f" % t_v if t_v else "-", float((got - ref).abs().max() / ref.abs().max())), flush=True)


def f2m_table(B, iters):
    """few -> many on the matrix cores (ipsr_conv_thin_f2m_mfma) against the vector-ALU kernel and MIOpen under autocast."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(9)
    print("\nfew -> many (batch %d)                              | MFMA thin ms | vector-ALU thin ms | MIOpen autocast ms" % B)
    cases = [("VGG conv1_1   conv  3 -> 64  k3 s1 @256 +bias,ReLU", ops.CONV_FWD, 3, 64, 256, 3, 1, torch.float32, True),
             ("netG first    conv  6 -> 64  k3 s1 @256", ops.CONV_FWD, 6, 64, 256, 3, 1, torch.float32, False),
             ("netD / netP   conv  3 -> 64  k4 s2 @256", ops.CONV_FWD, 3, 64, 256, 4, 2, torch.float32, False),
             ("netG last     convT 128 -> 3 k3 s1 @256 input gradient", ops.CONVT_BWD_DATA, 128, 3, 256, 3, 1, torch.bfloat16, False)]
    for name, op, Cin, Cout, S, k, st, idt, fused in cases:
        tr = op == ops.CONVT_BWD_DATA
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=dev, generator=g) * 0.1
        bias = torch.randn(Cout, device=dev, generator=g) * 0.1
        inp = (torch.rand(B, Cout if tr else Cin, S, S, device=dev, generator=g) * 2 - 1).to(idt)
        xb = (torch.rand(B, Cin, S, S, device=dev, generator=g)).to(torch.bfloat16) if tr else None
        wb = w.to(torch.bfloat16)

        def mfma():
            return ops.conv_thin_f2m_mfma(op, inp, w, (B, Cin, S, S), Cout, k, st, bias=bias if fused else None, relu=fused)

        def valu():
            return ops.conv3x3_thin(op, inp, w, (B, Cin, S, S), Cout, bias=bias if fused else None, relu=fused, out_dtype=torch.bfloat16)

        def miopen():
            if tr:
                return torch.ops.aten.convolution_backward(inp, xb, wb, None, [st, st], [1, 1], [1, 1], True, [0, 0], 1, [True, False, False])[0]
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = F.conv2d(inp, w, None, st, 1)
            return ops.bias_act_(y, bias, "relu") if fused else y
        t_m = timeit(mfma, iters)
        t_v = timeit(valu, iters) if k == 3 else None
        t_mi = timeit(miopen, iters)
        print("%-60s | %12.4f | %18s | %18.4f" % (name, t_m, "%.4f" % t_v if t_v else "-", t_mi), flush=True)


if __name__ == "__main__":
    main()
    wrw_table(16, 30)
    f2m_table(16, 30)
