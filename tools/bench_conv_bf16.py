#!/usr/bin/env python3
"""Direct bf16 3x3 convolution (csrc/conv_bf16.hip) per layer: error against an fp64 convolution of the bf16-rounded operands, time
against MIOpen's bf16 convolution (incl. its layout transposes and weight cast) and against the split-bf16 Winograd engine.

    python tools/bench_conv_bf16.py [--batch 16]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402,F401  (private MIOpen db copy)
from deepinpainting_amd import ops  # noqa: E402

LAYERS = [  # (kind, Cin, H, Cout)
    ("conv", 512, 32, 512), ("conv", 256, 32, 512), ("conv", 256, 64, 256), ("conv", 128, 64, 256), ("conv", 128, 128, 128), ("conv", 64, 128, 128),
    ("conv", 512, 16, 512), ("convT", 1024, 32, 256), ("convT", 512, 64, 128), ("convT", 256, 128, 64), ("convT", 1024, 16, 512),
]


def timeit(fn, iters=20):
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
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--only-s2", action="store_true")
    a = ap.parse_args()
    B = a.batch
    g = torch.Generator(device="cuda").manual_seed(5)
    print("layer (batch %d)         | pass  | direct bf16   ms     TF | MIOpen bf16  ms     TF | split-wino ms | err vs fp64(bf16 operands)" % B)
    for kind, Cin, H, Cout in ([] if a.only_s2 else LAYERS):
        tr = kind == "convT"
        wshape = (Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3)
        w = torch.randn(wshape, device="cuda", generator=g) * (1.0 / (3.0 * (Cin ** 0.5)))
        x = torch.randn(B, Cin, H, H, device="cuda", generator=g).to(torch.bfloat16)
        dy = torch.randn(B, Cout, H, H, device="cuda", generator=g).to(torch.bfloat16)
        wb = w.to(torch.bfloat16)
        fop, bop = (ops.CONVT_FWD, ops.CONVT_BWD_DATA) if tr else (ops.CONV_FWD, ops.CONV_BWD_DATA)
        flops = 2.0 * 9 * Cin * Cout * B * H * H
        for name, op, inp in (("fwd", fop, x), ("bwdD", bop, dy)):
            if not ops.conv3x3_bf16_supported(op, B, Cin, H, H, Cout):
                print("%-5s %4d->%-4d @%-3d   | %-5s | unsupported" % (kind, Cin, Cout, H, name))
                continue
            t_d = timeit(lambda: ops.conv3x3_bf16(op, inp, w, (B, Cin, H, H), Cout))
            if name == "fwd":
                mi = (lambda: F.conv_transpose2d(x, w.to(torch.bfloat16), None, 1, 1)) if tr else (lambda: F.conv2d(x, w.to(torch.bfloat16), None, 1, 1))
            else:
                mi = lambda: torch.ops.aten.convolution_backward(dy, x, w.to(torch.bfloat16), None, [1, 1], [1, 1], [1, 1], tr, [0, 0], 1, [True, False, False])
            t_m = timeit(mi)
            try:
                t_w = timeit(lambda: ops.conv3x3_winograd(op, inp, w, (B, Cin, H, H), Cout, math="bf16x3", out_dtype=torch.bfloat16))
            except Exception:
                t_w = float("nan")
            err = float("nan")
            if a.check:
                got = ops.conv3x3_bf16(op, inp, w, (B, Cin, H, H), Cout, out_dtype=torch.float32)[:2].double()
                xd, wd, dyd = x[:2].double(), wb.double(), dy[:2].double()
                if name == "fwd":
                    ref = F.conv_transpose2d(xd, wd, None, 1, 1) if tr else F.conv2d(xd, wd, None, 1, 1)
                else:
                    ref = F.conv2d(dyd, wd, None, 1, 1) if tr else F.conv_transpose2d(dyd, wd, None, 1, 1)
                err = float((got - ref).abs().max() / ref.abs().max())
            print("%-5s %4d->%-4d @%-3d   | %-5s | %10.4f %7.1f | %10.4f %7.1f | %10.4f    | %.2e" %
                  (kind, Cin, Cout, H, name, t_d, flops / t_d / 1e9, t_m, flops / t_m / 1e9, t_w, err), flush=True)
        if ops.conv3x3_bf16_wrw_supported(tr, B, Cin, H, H, Cout):
            t_d = timeit(lambda: ops.conv3x3_bf16_wrw(tr, x, dy, Cout))
            t_m = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w.to(torch.bfloat16), None, [1, 1], [1, 1], [1, 1], tr, [0, 0], 1, [False, True, False])[1].float())
            try:
                t_w = timeit(lambda: ops.conv3x3_winograd_wrw(tr, x, dy, Cout, math="bf16x3"))
            except Exception:
                t_w = float("nan")
            err = float("nan")
            if a.check:
                got = ops.conv3x3_bf16_wrw(tr, x, dy, Cout).double()
                ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], tr, [0, 0], 1, [False, True, False])[1]
                err = float((got - ref).abs().max() / ref.abs().max())
            print("%-5s %4d->%-4d @%-3d   | %-5s | %10.4f %7.1f | %10.4f %7.1f | %10.4f    | %.2e" %
                  (kind, Cin, Cout, H, "wrw", t_d, flops / t_d / 1e9, t_m, flops / t_m / 1e9, t_w, err), flush=True)


    # ---- 4x4 stride-2 pad-1 family: Conv2d (fine -> coarse forward) and ConvTranspose2d (coarse -> fine forward), + their input gradients
    print()
    S2 = [("conv", 64, 128, 128), ("conv", 128, 64, 256), ("conv", 256, 32, 512), ("conv", 512, 16, 512),
          ("convT", 512, 16, 512), ("convT", 1024, 16, 256), ("convT", 512, 32, 128), ("convT", 256, 32, 256), ("convT", 256, 64, 64), ("convT", 128, 64, 128),
          ("convT", 64, 128, 64), ("convT", 128, 128, 3)]
    for kind, Cin, H, Cout in S2:
        tr = kind == "convT"
        Kc, Cf = (Cin, Cout) if tr else (Cout, Cin)           # weight [Kc][Cf][4][4] in both modules
        nh = H if tr else H // 2
        w = torch.randn(Kc, Cf, 4, 4, device="cuda", generator=g) * (1.0 / (4.0 * (Cin ** 0.5)))
        fine = torch.randn(B, Cf, 2 * nh, 2 * nh, device="cuda", generator=g).to(torch.bfloat16)
        coarse = torch.randn(B, Kc, nh, nh, device="cuda", generator=g).to(torch.bfloat16)
        wb = w.to(torch.bfloat16)
        flops = 2.0 * 16 * Kc * Cf * B * nh * nh
        x, dy = (coarse, fine) if tr else (fine, coarse)
        for name in ("fwd", "bwdD"):
            mode = (ops.S2_COARSE_TO_FINE if name == "fwd" else ops.S2_FINE_TO_COARSE) if tr else (ops.S2_FINE_TO_COARSE if name == "fwd" else ops.S2_COARSE_TO_FINE)
            inp = x if name == "fwd" else dy
            if not ops.conv4x4s2_bf16_supported(mode, B, Kc, Cf, nh, nh):
                print("%-5s %4d->%-4d @%-3d k4s2 | %-5s | unsupported" % (kind, Cin, Cout, H, name))
                continue
            t_d = timeit(lambda: ops.conv4x4s2_bf16(mode, inp, w, B, Kc, Cf, nh, nh))
            if name == "fwd":
                mi = (lambda: F.conv_transpose2d(x, w.to(torch.bfloat16), None, 2, 1)) if tr else (lambda: F.conv2d(x, w.to(torch.bfloat16), None, 2, 1))
            else:
                mi = lambda: torch.ops.aten.convolution_backward(dy, x, w.to(torch.bfloat16), None, [2, 2], [1, 1], [1, 1], tr, [0, 0], 1, [True, False, False])
            t_m = timeit(mi)
            try:
                t_w = timeit(lambda: ops.conv4x4s2_winograd(mode, inp, w, B, Kc, Cf, nh, nh, math="bf16x3", out_dtype=torch.bfloat16))
            except Exception:
                t_w = float("nan")
            err = float("nan")
            if a.check:
                got = ops.conv4x4s2_bf16(mode, inp, w, B, Kc, Cf, nh, nh, out_dtype=torch.float32)[:2].double()
                if mode == ops.S2_FINE_TO_COARSE:
                    ref = F.conv2d(fine[:2].double(), wb.double(), None, 2, 1)
                else:
                    ref = F.conv_transpose2d(coarse[:2].double(), wb.double(), None, 2, 1)
                err = float((got - ref).abs().max() / ref.abs().max())
            print("%-5s %4d->%-4d @%-3d k4s2 | %-5s | %10.4f %7.1f | %10.4f %7.1f | %10.4f    | %.2e" %
                  (kind, Cin, Cout, H, name, t_d, flops / t_d / 1e9, t_m, flops / t_m / 1e9, t_w, err), flush=True)
        if ops.conv4x4s2_bf16_wrw_supported(B, Kc, Cf, nh, nh):
            t_d = timeit(lambda: ops.conv4x4s2_bf16_wrw(fine, coarse, B, Kc, Cf, nh, nh))
            t_m = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w.to(torch.bfloat16), None, [2, 2], [1, 1], [1, 1], tr, [0, 0], 1, [False, True, False])[1].float())
            try:
                t_w = timeit(lambda: ops.conv4x4s2_winograd(ops.S2_WEIGHT_GRAD, fine, coarse, B, Kc, Cf, nh, nh, math="bf16x3"))
            except Exception:
                t_w = float("nan")
            err = float("nan")
            if a.check:
                got = ops.conv4x4s2_bf16_wrw(fine, coarse, B, Kc, Cf, nh, nh).double()
                ref = torch.ops.aten.convolution_backward(coarse.double(), fine.double(), w.double(), None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
                err = float((got - ref).abs().max() / ref.abs().max())
            print("%-5s %4d->%-4d @%-3d k4s2 | %-5s | %10.4f %7.1f | %10.4f %7.1f | %10.4f    | %.2e" %
                  (kind, Cin, Cout, H, "wrw", t_d, flops / t_d / 1e9, t_m, flops / t_m / 1e9, t_w, err), flush=True)


if __name__ == "__main__":
    main()
