#!/usr/bin/env python3
"""Per-shape check + timing of the hand-written implicit-GEMM convolutions (csrc/conv_gemm.hip) against MIOpen, for every
distinct Conv2d / ConvTranspose2d geometry of the training step at BASELINE config 2 (batch 8, 256x256, fp32).

    python tools/bench_hipconv.py [--batch 8] [--quick] > gpurun_out/hipconv.txt

For each shape and op (forward, backward-data): max |err| of the HIP kernel and of MIOpen against an fp64 CPU convolution of
sample 0, and the time of both (median of 10, HIP events).  `use` marks where the HIP kernel is the faster one.
"""
import argparse
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401  (private MIOpen db copy)
from deepinpainting_amd import ops

# (kind, Cin, H, Cout, k, stride, pad, dil) — the step's layers (profiles/r01_conv_layer_table.txt)
SHAPES = [
    ("conv", 512, 32, 512, 3, 1, 1, 1), ("conv", 256, 64, 256, 3, 1, 1, 1), ("conv", 64, 256, 64, 3, 1, 1, 1),
    ("conv", 128, 128, 128, 3, 1, 1, 1), ("conv", 64, 128, 128, 3, 1, 1, 1), ("conv", 128, 64, 256, 3, 1, 1, 1),
    ("conv", 256, 32, 512, 3, 1, 1, 1), ("conv", 256, 32, 512, 4, 1, 1, 1),
    ("conv", 256, 32, 512, 4, 2, 1, 1), ("conv", 64, 128, 128, 4, 2, 1, 1), ("conv", 128, 64, 256, 4, 2, 1, 1),
    ("conv", 512, 16, 512, 4, 2, 1, 1), ("conv", 512, 8, 512, 4, 2, 1, 1), ("conv", 512, 4, 512, 4, 2, 1, 1),
    ("conv", 512, 2, 512, 4, 2, 1, 1),
    ("conv", 64, 256, 64, 4, 2, 3, 2), ("conv", 128, 128, 128, 4, 2, 3, 2), ("conv", 256, 64, 256, 4, 2, 3, 2),
    ("conv", 512, 32, 512, 4, 2, 3, 2), ("conv", 512, 16, 512, 4, 2, 3, 2), ("conv", 512, 8, 512, 4, 2, 3, 2),
    ("conv", 512, 4, 512, 4, 2, 3, 2), ("conv", 512, 2, 512, 4, 2, 3, 2),
    ("conv", 512, 16, 512, 3, 1, 1, 1), ("conv", 512, 8, 512, 3, 1, 1, 1), ("conv", 512, 4, 512, 3, 1, 1, 1),
    ("conv", 512, 2, 512, 3, 1, 1, 1), ("conv", 6, 256, 64, 3, 1, 1, 1),
    ("convT", 256, 128, 64, 3, 1, 1, 1), ("convT", 512, 64, 128, 3, 1, 1, 1), ("convT", 1024, 32, 256, 3, 1, 1, 1),
    ("convT", 1024, 16, 512, 3, 1, 1, 1), ("convT", 1024, 8, 512, 3, 1, 1, 1), ("convT", 1024, 4, 512, 3, 1, 1, 1),
    ("convT", 1024, 2, 512, 3, 1, 1, 1),
    ("convT", 64, 128, 64, 4, 2, 1, 1), ("convT", 1024, 16, 256, 4, 2, 1, 1), ("convT", 128, 64, 128, 4, 2, 1, 1),
    ("convT", 512, 16, 512, 4, 2, 1, 1), ("convT", 256, 64, 64, 4, 2, 1, 1), ("convT", 512, 32, 128, 4, 2, 1, 1),
    ("convT", 256, 32, 256, 4, 2, 1, 1), ("convT", 1024, 8, 512, 4, 2, 1, 1), ("convT", 512, 8, 512, 4, 2, 1, 1),
    ("convT", 1024, 4, 512, 4, 2, 1, 1), ("convT", 512, 4, 512, 4, 2, 1, 1), ("convT", 1024, 2, 512, 4, 2, 1, 1),
    ("convT", 512, 2, 512, 4, 2, 1, 1), ("convT", 512, 1, 512, 4, 2, 1, 1),
]
K3 = [s_ for s_ in SHAPES if s_[4] == 3]
QUICK = [("conv", 512, 32, 512, 3, 1, 1, 1), ("conv", 256, 32, 512, 4, 2, 1, 1), ("conv", 512, 32, 512, 4, 2, 3, 2),
         ("conv", 512, 8, 512, 4, 2, 1, 1), ("convT", 1024, 32, 256, 3, 1, 1, 1), ("convT", 512, 32, 128, 4, 2, 1, 1),
         ("convT", 512, 4, 512, 4, 2, 1, 1)]


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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--k3", action="store_true", help="only the 3x3 stride-1 shapes (the Winograd candidates)")
    ap.add_argument("--k4s1", action="store_true", help="only netD's 4x4 stride-1 convolution")
    ap.add_argument("--small", action="store_true", help="only the small-map layers (inner U-Net levels, netF) on ipsr_conv_smallmap")
    ap.add_argument("--k4s2", action="store_true", help="only the 4x4 stride-2 pad-1 layers (polyphase Winograd F(5x5,2x2))")
    args = ap.parse_args()
    B = args.batch
    torch.manual_seed(0)
    print("%-5s %-16s %-14s %-7s | %-4s %9s %9s %7s %7s | %9s %9s  %s" % ("kind", "input", "weight", "s/p/d", "op", "hip ms", "miopen ms", "hip TF", "mio TF", "hip err", "mio err", "use"))
    tot_h = tot_m = tot_best = 0.0
    for kind, Cin, H, Cout, k, st, pad, dil in ([] if (args.k4s1 or args.k4s2 or args.small) else QUICK if args.quick else (K3 if args.k3 else SHAPES)):
        tr = kind == "convT"
        x = torch.randn(B, Cin, H, H, device="cuda")
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device="cuda") * 0.05
        f = (lambda a, ww: F.conv_transpose2d(a, ww, None, st, pad, 0, 1, dil)) if tr else (lambda a, ww: F.conv2d(a, ww, None, st, pad, dil))
        with torch.no_grad():
            y = f(x, w)
        Ho = y.shape[2]
        dy = torch.randn_like(y)
        flops = 2.0 * B * Cin * Cout * k * k * (H * H if tr else Ho * Ho)
        cb = torch.ops.aten.convolution_backward
        cargs = (dy, x, w, None, [st, st], [pad, pad], [dil, dil], tr, [0, 0], 1)
        for opname, op in (("fwd", ops.CONVT_FWD if tr else ops.CONV_FWD), ("bwdD", ops.CONVT_BWD_DATA if tr else ops.CONV_BWD_DATA)):
            fwd = opname == "fwd"
            mio = (lambda: f(x, w)) if fwd else (lambda: cb(*cargs, [True, False, False])[0])
            if not ops.conv2d_supported(op, B, Cin, H, H, Cout, k, st, pad, dil):
                with torch.no_grad():
                    tm = timed(mio)
                print("%-5s %-16s %-14s %-7s | %-4s %9s %9.4f %7s %7.1f | unsupported -> MIOpen" % (
                    kind, "%dx%dx%dx%d" % (B, Cin, H, H), "x".join(map(str, w.shape)), "%d/%d/%d" % (st, pad, dil), opname, "-", tm, "-", flops / tm / 1e9))
                tot_m += tm; tot_best += tm; tot_h += tm
                continue
            hip = (lambda: ops.conv2d(op, x if fwd else dy, w, (B, Cin, H, H), Cout, k, st, pad, dil))
            with torch.no_grad():
                got, ref32 = hip(), mio()
                eh = em = float("nan")
                if not args.no_check:
                    xd, wd, dyd = x[:1].double().cpu().requires_grad_(True), w.double().cpu(), dy[:1].double().cpu()
                    with torch.enable_grad():
                        yd = f(xd, wd)
                        want = yd.detach() if fwd else torch.autograd.grad(yd, xd, dyd)[0]
                    scale = float(want.abs().max())
                    eh = float((got[:1].double().cpu() - want).abs().max()) / scale
                    em = float((ref32[:1].double().cpu() - want).abs().max()) / scale
                    rest = float((got - ref32).abs().max()) / scale           # the other samples against MIOpen
                    eh = max(eh, rest if rest > 1e-4 else 0.0)
                th, tm = timed(hip), timed(mio)
            tw, ew = float("inf"), float("nan")
            if k == 3 and st == 1 and pad == 1 and dil == 1 and ops.winograd_supported(op, B, Cin, H, H, Cout):
                wino = (lambda: ops.conv3x3_winograd(op, x if fwd else dy, w, (B, Cin, H, H), Cout))
                gw = wino()
                if not args.no_check:
                    ew = max(float((gw[:1].double().cpu() - want).abs().max()) / scale, float((gw - ref32).abs().max()) / scale)
                tw = timed(wino)
            best = min(th, tm, tw)
            tot_h += min(th, tw); tot_m += tm; tot_best += best
            print("%-5s %-16s %-14s %-7s | %-4s %9.4f %9.4f %7.1f %7.1f | %9.2e %9.2e  %-4s | wino %9.4f ms %7.1f TF err %9.2e" % (
                kind, "%dx%dx%dx%d" % (B, Cin, H, H), "x".join(map(str, w.shape)), "%d/%d/%d" % (st, pad, dil), opname, th, tm,
                flops / th / 1e9, flops / tm / 1e9, eh, em, "WINO" if best == tw else ("HIP" if best == th else ""),
                tw, flops / tw / 1e9 if tw < 1e9 else 0.0, ew), flush=True)
    if args.k3:
        print("\nweight gradient, Winograd F(3x3,4x4) vs MIOpen:")
        for kind, Cin, H, Cout, k, st, pad, dil in K3:
            tr = kind == "convT"
            if min(Cin, Cout) < 16:
                continue
            x = torch.randn(B, Cin, H, H, device="cuda")
            w = torch.randn((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3), device="cuda") * 0.05
            dy = torch.randn(B, Cout, H, H, device="cuda")
            cb = torch.ops.aten.convolution_backward
            cargs = (dy, x, w, None, [1, 1], [1, 1], [1, 1], tr, [0, 0], 1)
            with torch.no_grad():
                mio = lambda: cb(*cargs, [False, True, False])[1]
                hipw = lambda: ops.conv3x3_winograd_wrw(tr, x, dy, Cout)
                a, b2 = hipw(), mio()
                err = float((a - b2).abs().max() / b2.abs().max())
                tw, tm = timed(hipw), timed(mio)
            flops = 2.0 * B * Cin * Cout * 9 * H * H
            print("%-5s %-16s %-14s | wrw wino %9.4f ms %7.1f TF   miopen %9.4f ms %7.1f TF   |wino - miopen| %9.2e  %s" % (
                kind, "%dx%dx%dx%d" % (B, Cin, H, H), "x".join(map(str, w.shape)), tw, flops / tw / 1e9, tm, flops / tm / 1e9, err,
                "WINO" if tw < tm else ""), flush=True)
    if args.k3:
        print("\ndilated 4x4 stride-2 (netG down convolutions), Winograd F(3x3,4x4) vs MIOpen:")
        for Cin, H in ((64, 256), (128, 128), (256, 64), (512, 32), (512, 16), (512, 8)):
            Cout = Cin
            x = torch.randn(B, Cin, H, H, device="cuda")
            w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
            dy = torch.randn(B, Cout, H // 2, H // 2, device="cuda")
            cb = torch.ops.aten.convolution_backward
            cargs = (dy, x, w, None, [2, 2], [3, 3], [2, 2], False, [0, 0], 1)
            flops = 2.0 * B * Cin * Cout * 16 * (H // 2) ** 2
            with torch.no_grad():
                for name, hipf, miof in (("fwd", lambda: ops.conv4x4_dilated_winograd(0, x, w, (B, Cin, H, H), Cout), lambda: F.conv2d(x, w, None, 2, 3, 2)),
                                         ("bwdD", lambda: ops.conv4x4_dilated_winograd(1, dy, w, (B, Cin, H, H), Cout), lambda: cb(*cargs, [True, False, False])[0]),
                                         ("wrw", lambda: ops.conv4x4_dilated_winograd(2, x, dy, (B, Cin, H, H), Cout), lambda: cb(*cargs, [False, True, False])[1])):
                    a, b2 = hipf(), miof()
                    err = float((a - b2).abs().max() / b2.abs().max())
                    tw, tm = timed(hipf), timed(miof)
                    print("dil   %-16s %-14s | %-4s wino %9.4f ms %7.1f TF   miopen %9.4f ms %7.1f TF   |diff| %9.2e  %s" % (
                        "%dx%dx%dx%d" % (B, Cin, H, H), "x".join(map(str, w.shape)), name, tw, flops / tw / 1e9, tm, flops / tm / 1e9, err,
                        "WINO" if tw < tm else ""), flush=True)
    if args.small:
        print("\nsmall maps (inner U-Net levels, netF): ipsr_conv_smallmap vs MIOpen")
        cb = torch.ops.aten.convolution_backward
        SMALL = [("conv", 512, 512, 8, 3, 1, 1, 1), ("conv", 512, 512, 4, 3, 1, 1, 1), ("conv", 512, 512, 2, 3, 1, 1, 1),
                 ("conv", 512, 512, 16, 4, 2, 3, 2), ("conv", 512, 512, 8, 4, 2, 3, 2), ("conv", 512, 512, 4, 4, 2, 3, 2), ("conv", 512, 512, 2, 4, 2, 3, 2),
                 ("conv", 512, 512, 16, 4, 2, 1, 1), ("conv", 512, 512, 8, 4, 2, 1, 1), ("conv", 512, 512, 4, 4, 2, 1, 1), ("conv", 512, 512, 2, 4, 2, 1, 1),
                 ("convT", 1024, 512, 8, 3, 1, 1, 1), ("convT", 1024, 512, 4, 3, 1, 1, 1), ("convT", 1024, 512, 2, 3, 1, 1, 1),
                 ("convT", 512, 512, 1, 4, 2, 1, 1), ("convT", 512, 512, 2, 4, 2, 1, 1), ("convT", 512, 512, 4, 4, 2, 1, 1), ("convT", 512, 512, 8, 4, 2, 1, 1),
                 ("convT", 1024, 512, 2, 4, 2, 1, 1), ("convT", 1024, 512, 4, 4, 2, 1, 1), ("convT", 1024, 512, 8, 4, 2, 1, 1)]
        tot_h = tot_m = 0.0
        for kind, Ci, Co, H, k, st, pad, dil in SMALL:
            tr = kind == "convT"
            x = torch.randn(B, Ci, H, H, device="cuda")
            w = torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), device="cuda") * 0.05
            with torch.no_grad():
                y0 = F.conv_transpose2d(x, w, None, st, pad, 0, 1, dil) if tr else F.conv2d(x, w, None, st, pad, dil)
            Hy = y0.shape[2]
            dy = torch.randn_like(y0)
            cargs = (dy, x, w, None, [st, st], [pad, pad], [dil, dil], tr, [0, 0], 1)
            flops = 2.0 * B * Ci * Co * k * k * (H * H if tr else Hy * Hy)
            if tr:
                geo = (B, Ci, Co, H, H, Hy, Hy, k, st, pad, dil)
                rows = (("fwd", lambda: ops.conv_smallmap(ops.SM_DATA, x, w, *geo), lambda: F.conv_transpose2d(x, w, None, st, pad, 0, 1, dil)),
                        ("wrw", lambda: ops.conv_smallmap(ops.SM_WRW, x, dy, *geo), lambda: cb(*cargs, [False, True, False])[1]))
            else:
                geo = (B, Co, Ci, Hy, Hy, H, H, k, st, pad, dil)
                rows = (("bwdD", lambda: ops.conv_smallmap(ops.SM_DATA, dy, w, *geo), lambda: cb(*cargs, [True, False, False])[0]),
                        ("wrw", lambda: ops.conv_smallmap(ops.SM_WRW, dy, x, *geo), lambda: cb(*cargs, [False, True, False])[1]))
            with torch.no_grad():
                for name, hipf, miof in rows:
                    r1, r2 = hipf(), miof()
                    err = float((r1 - r2).abs().max() / r2.abs().max())
                    tw, tm = timed(hipf), timed(miof)
                    tot_h += tw; tot_m += tm
                    print("small %-5s %4d->%4d @%2d k%d s%d p%d d%d | %-4s here %9.4f ms %6.1f TF   miopen %9.4f ms %6.1f TF   |diff| %9.2e  %s" % (
                        kind, Ci, Co, H, k, st, pad, dil, name, tw, flops / tw / 1e9, tm, flops / tm / 1e9, err, "HERE" if tw < tm else ""), flush=True)
        print("small-map total (one call each): here %.3f ms, miopen %.3f ms" % (tot_h, tot_m))
    if args.k4s2:
        print("\n4x4 stride-2 pad-1 (Conv2d of netP/netD/netF, ConvTranspose2d of netP/netG), polyphase Winograd F(5x5,2x2) vs MIOpen:")
        cb = torch.ops.aten.convolution_backward
        LAYERS = [("conv", 128, 64, 64), ("conv", 256, 128, 32), ("conv", 512, 256, 16), ("conv", 512, 512, 8), ("conv", 512, 512, 4),
                  ("convT", 64, 64, 128), ("convT", 128, 128, 64), ("convT", 256, 256, 32), ("convT", 512, 512, 16), ("convT", 256, 64, 64),
                  ("convT", 512, 128, 32), ("convT", 1024, 256, 16), ("convT", 1024, 512, 8), ("convT", 512, 512, 8)]
        for kind, Kc, Cf, n in LAYERS:
            fine = torch.randn(B, Cf, 2 * n, 2 * n, device="cuda")
            coarse = torch.randn(B, Kc, n, n, device="cuda")
            w = torch.randn(Kc, Cf, 4, 4, device="cuda") * 0.05
            flops = 2.0 * B * Kc * Cf * 16 * n * n
            tr = kind == "convT"
            if not tr:      # Conv2d: x = fine, dy = coarse
                cargs = (coarse, fine, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1)
                rows = (("fwd", 0, fine, w, lambda: F.conv2d(fine, w, None, 2, 1)),
                        ("bwdD", 1, coarse, w, lambda: cb(*cargs, [True, False, False])[0]),
                        ("wrw", 2, fine, coarse, lambda: cb(*cargs, [False, True, False])[1]))
            else:           # ConvTranspose2d: x = coarse, dy = fine
                cargs = (fine, coarse, w, None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1)
                rows = (("fwd", 1, coarse, w, lambda: F.conv_transpose2d(coarse, w, None, 2, 1)),
                        ("bwdD", 0, fine, w, lambda: cb(*cargs, [True, False, False])[0]),
                        ("wrw", 2, fine, coarse, lambda: cb(*cargs, [False, True, False])[1]))
            with torch.no_grad():
                for name, mode, a_, b_, miof in rows:
                    if not ops.s2_winograd_supported(mode, B, Kc, Cf, n, n):
                        continue
                    hipf = lambda: ops.conv4x4s2_winograd(mode, a_, b_, B, Kc, Cf, n, n)
                    r1, r2 = hipf(), miof()
                    err = float((r1 - r2).abs().max() / r2.abs().max())
                    tw, tm = timed(hipf), timed(miof)
                    print("k4s2  %-5s Kc=%4d Cf=%4d n=%3d | %-4s wino %9.4f ms %7.1f TF   miopen %9.4f ms %7.1f TF   |diff| %9.2e  %s" % (
                        kind, Kc, Cf, n, name, tw, flops / tw / 1e9, tm, flops / tm / 1e9, err, "WINO" if tw < tm else ""), flush=True)
    if args.k3 or args.k4s1:
        print("\n4x4 stride-1 pad-1 (netD's fourth convolution), Winograd F(3x3,4x4) vs MIOpen:")
        for Cin, H, Cout in ((256, 32, 512), (128, 64, 256), (512, 16, 512)):
            x = torch.randn(B, Cin, H, H, device="cuda")
            w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
            dy = torch.randn(B, Cout, H - 1, H - 1, device="cuda")
            cb = torch.ops.aten.convolution_backward
            cargs = (dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
            flops = 2.0 * B * Cin * Cout * 16 * (H - 1) ** 2
            G1 = ops.GEOM_K4_S1_P1
            with torch.no_grad():
                for name, hipf, miof in (("fwd", lambda: ops.conv4x4_dilated_winograd(0, x, w, (B, Cin, H, H), Cout, geom=G1), lambda: F.conv2d(x, w, None, 1, 1, 1)),
                                         ("bwdD", lambda: ops.conv4x4_dilated_winograd(1, dy, w, (B, Cin, H, H), Cout, geom=G1), lambda: cb(*cargs, [True, False, False])[0]),
                                         ("wrw", lambda: ops.conv4x4_dilated_winograd(2, x, dy, (B, Cin, H, H), Cout, geom=G1), lambda: cb(*cargs, [False, True, False])[1])):
                    a, b2 = hipf(), miof()
                    err = float((a - b2).abs().max() / b2.abs().max())
                    tw, tm = timed(hipf), timed(miof)
                    print("k4s1  %-16s %-14s | %-4s wino %9.4f ms %7.1f TF   miopen %9.4f ms %7.1f TF   |diff| %9.2e  %s" % (
                        "%dx%dx%dx%d" % (B, Cin, H, H), "x".join(map(str, w.shape)), name, tw, flops / tw / 1e9, tm, flops / tm / 1e9, err,
                        "WINO" if tw < tm else ""), flush=True)
    print("sum over shapes (one call each): hip %.3f ms, miopen %.3f ms, best-of %.3f ms" % (tot_h, tot_m, tot_best))


if __name__ == "__main__":
    main()
