#!/usr/bin/env python3
"""Which engine runs which convolution call of one training step (BASELINE config 2: batch 8, 256x256, fp32; `--dtype bf16 --batch 16`:
config 5), and what the calls
that stay on MIOpen cost.  Every call of models/hipconv.py (the U-Nets and discriminators; the frozen VGG16 goes through
models/vgg16.py's own path and is listed from the dispatcher) reports (pass, engine, geometry) through hipconv._check_hook; the
MIOpen ones are then timed stand-alone on tensors of the same shapes (median of 10, HIP events, transposes included).

    python tools/engine_map.py > gpurun_out/engine_map.txt
"""
import collections
import contextlib
import io
import os
import statistics
import sys
import tempfile

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401,E402  (private MIOpen find-db copy, as the benchmark runs)
from deepinpainting_amd.models import hipconv  # noqa: E402
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402


def _i(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


def time_ms(fn, n=10):
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


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32", help="bf16 = BASELINE config 5 (convolutions under bf16 autocast)")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    B, bf = a.batch, a.dtype == "bf16"
    act = torch.bfloat16 if bf else torch.float32
    opt = Option(gpu_ids=[0], batchSize=B, use_dropout=True, quiet=True, allow_random_vgg=True, amp_bf16=bf, checkpoints_dir=tempfile.mkdtemp())
    torch.manual_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(opt)
    g = torch.Generator(device="cuda").manual_seed(21)
    img = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    ref = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    calls = collections.Counter()

    def hook(kind, engine, geom, operands, result):
        transposed, k, stride, pad, dil, Cout = geom
        x = operands[0] if kind == "forward" else operands[1]
        w = operands[-1]
        calls[(kind, engine, transposed, tuple(x.shape), tuple(w.shape), stride, pad, dil)] += 1

    # layers whose three passes ALL stay on MIOpen never enter _HipConv (conv_nobias / vgg16.py call F.conv2d directly): count them
    # at torch.nn.functional (forward here; autograd then runs MIOpen's two backward passes where a gradient is needed)
    plain = collections.Counter()
    real_conv2d, real_convT = F.conv2d, F.conv_transpose2d

    def rec_conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if plain_on[0] and x.is_cuda:
            plain[("conv", tuple(x.shape), tuple(w.shape), _i(stride), _i(padding), _i(dilation), bool(x.requires_grad), bool(w.requires_grad))] += 1
        return real_conv2d(x, w, b, stride, padding, dilation, groups)

    def rec_convT(x, w, b=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        if plain_on[0] and x.is_cuda:
            plain[("convT", tuple(x.shape), tuple(w.shape), _i(stride), _i(padding), _i(dilation), bool(x.requires_grad), bool(w.requires_grad))] += 1
        return real_convT(x, w, b, stride, padding, output_padding, groups, dilation)

    plain_on = [False]
    F.conv2d, F.conv_transpose2d = rec_conv2d, rec_convT
    for step in range(2):
        hipconv._check_hook = hook if step == 1 else None
        plain_on[0] = step == 1
        try:
            m.set_input(img, mask, ref)
            m.set_ref_latent()
            m.set_gt_latent()
            m.optimize_parameters()
        finally:
            hipconv._check_hook = None
            plain_on[0] = False
    F.conv2d, F.conv_transpose2d = real_conv2d, real_convT
    torch.cuda.synchronize()
    by_engine = collections.Counter()
    for (kind, engine, *_), n in calls.items():
        by_engine[(kind, engine)] += n
    print("convolution calls of one step through models/hipconv.py: %d" % sum(calls.values()))
    for (kind, engine), n in sorted(by_engine.items()):
        print("  %-12s %-9s %3d calls" % (kind, engine, n))
    print("\ncalls that stay on MIOpen (stand-alone time per call, transposes included):")
    total = 0.0
    rows = []
    for (kind, engine, tr, xs, ws, st, pd, dl), n in calls.items():
        if engine != "miopen":
            continue
        x = torch.randn(xs, device="cuda").to(act)
        w = torch.randn(ws, device="cuda") * 0.05                   # fp32 parameter: under autocast every call casts it (timed, as in the step)
        f = (lambda a, b: F.conv_transpose2d(a, b.to(act), None, st, pd, 0, 1, dl)) if tr else (lambda a, b: F.conv2d(a, b.to(act), None, st, pd, dl))
        if kind == "forward":
            ms = time_ms(lambda: f(x, w))
        else:
            dy = torch.randn_like(f(x, w))
            which = [kind == "input_grad", kind == "weight_grad", False]
            ms = time_ms(lambda: torch.ops.aten.convolution_backward(dy, x, w.to(act), None, [st, st], [pd, pd], [dl, dl], tr, [0, 0], 1, which))
        rows.append((ms * n, "  %-12s %-5s x %-18s w %-20s s%d p%d d%d   %d x %.3f ms" % (kind, "convT" if tr else "conv", "x".join(map(str, xs)),
                                                                                         "x".join(map(str, ws)), st, pd, dl, n, ms)))
        total += ms * n
    for _, line in sorted(rows, reverse=True):
        print(line)
    print("  total %.2f ms per step" % total)
    print("\nlayers with all passes on MIOpen (F.conv2d / F.conv_transpose2d called directly; forward + the backward passes autograd needs):")
    total2, rows = 0.0, []
    for (kind, xs, ws, st, pd, dl, gx, gw), n in plain.items():
        tr = kind == "convT"
        x = torch.randn(xs, device="cuda").to(act)
        w = torch.randn(ws, device="cuda") * 0.05
        f = (lambda a, b: real_convT(a, b.to(act), None, st, pd, 0, 1, dl)) if tr else (lambda a, b: real_conv2d(a, b.to(act), None, st, pd, dl))
        ms = [time_ms(lambda: f(x, w))]
        dy = torch.randn_like(f(x, w))
        for need, which in ((gx, [True, False, False]), (gw, [False, True, False])):
            ms.append(time_ms(lambda: torch.ops.aten.convolution_backward(dy, x, w.to(act), None, [st, st], [pd, pd], [dl, dl], tr, [0, 0], 1, which)) if need else 0.0)
        rows.append((sum(ms) * n, "  %-5s x %-18s w %-20s s%d p%d d%d   %d x (fwd %.3f + dx %.3f + dw %.3f ms)" % (
            kind, "x".join(map(str, xs)), "x".join(map(str, ws)), st, pd, dl, n, ms[0], ms[1], ms[2])))
        total2 += sum(ms) * n
    for _, line in sorted(rows, reverse=True):
        print(line)
    print("  total %.2f ms per step" % total2)


if __name__ == "__main__":
    main()
