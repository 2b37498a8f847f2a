#!/usr/bin/env python3
"""Eager training steps against the same steps replayed from one HIP graph (deepinpainting_amd/stepgraph.py): values and time.

    python tools/exp_stepgraph.py [--dtype bf16] [--batch 16] [--steps 4]
"""
import argparse
import contextlib
import io
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(args, tag):
    from deepinpainting_amd.options import Option
    from deepinpainting_amd.models.models import create_model
    opt = Option(gpu_ids=[0], batchSize=args.batch, use_dropout=not args.no_dropout, quiet=True, allow_random_vgg=True, amp_bf16=args.dtype == "bf16",
                 checkpoints_dir="/tmp/exp_stepgraph_" + tag)
    torch.manual_seed(1234 if args.bench_like else 77)
    with contextlib.redirect_stdout(io.StringIO()):
        return create_model(opt)


def data(args, i):
    if args.bench_like:
        import bench
        img, _, ref = bench.synthetic_batch(torch.device("cuda", 0), args.batch, 1234)
        return img, ref
    g = torch.Generator(device="cuda").manual_seed(100 + i)
    img = torch.rand(args.batch, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    ref = torch.rand(args.batch, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    return img, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--x", default="", help="comma list of bench.py look-alikes: db,bm,freeze,setdev,lib,init")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--time-steps", type=int, default=20)
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--pre", type=int, default=0, help="eager steps on the default stream before the first StepGraph call (as bench.py's warm-up)")
    ap.add_argument("--modes", default="eager,warm,graph")
    ap.add_argument("--bench-like", action="store_true", help="bench.py's seed (1234) and its one synthetic batch for every step")
    ap.add_argument("--host-churn", action="store_true", help="overwrite freed host memory between replays (a graph node that kept a host pointer would read it)")
    ap.add_argument("--dev-churn", action="store_true", help="allocate, fill and free device tensors of many sizes between replays")
    ap.add_argument("--quiet-steps", action="store_true", help="no host read between the compared steps either (losses read once after the last)")
    ap.add_argument("--events", action="store_true", help="record a HIP event on the launch stream after every step of the timing loop (bench.py does)")
    ap.add_argument("--no-sync", action="store_true", help="no host read of the losses inside the timing loop")
    ap.add_argument("--fast", action="store_true", help="MIOpen's default (non-deterministic) solvers: for the timing")
    args = ap.parse_args()
    xs = set(args.x.split(","))
    if "db" in xs:
        from deepinpainting_amd import use_shipped_miopen_db
        use_shipped_miopen_db()
    if "init" in xs:
        from deepinpainting_amd import dist as idist
        idist.init_distributed(backend="nccl")
    if "setdev" in xs:
        torch.cuda.set_device(0)
    if "bm" in xs:
        torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = not args.fast
    from deepinpainting_amd.stepgraph import StepGraph
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1

    out = {}
    for mode in args.modes.split(","):
        m = build(args, mode)
        if not args.bench_like:
            torch.cuda.manual_seed(999)
        sg = StepGraph(m, capture=mode == "graph") if mode != "eager" else None
        losses = []
        for i in range(args.pre):
            img, ref = data(args, 50 + i)
            m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()
            if args.bench_like:
                torch.cuda.synchronize()
        if args.bench_like and args.pre:
            import gc
            gc.collect()
            if "freeze" in xs:
                gc.freeze()
        if "lib" in xs:
            from deepinpainting_amd import _lib
            _lib.lib()
        for i in range(args.steps):
            img, ref = data(args, i)
            if sg is None:
                m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()
            else:
                sg.step(img, mask, ref)
            if not args.quiet_steps or i == args.steps - 1:
                e = m.get_current_errors()
                losses.append([e[k] for k in ("G_GAN", "G_L1", "D", "F")])
        w = torch.cat([p.detach().flatten() for net in (m.netG, m.netP, m.netD, m.netF) for p in net.parameters()]).clone()
        # time
        img, ref = data(args, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs = []
        for i in range(args.time_steps):
            if sg is None:
                m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()
            else:
                sg.step(img, mask, ref)
            if args.host_churn:
                junk = [bytes([0xFF]) * n for n in (64, 256, 1024, 4096, 65536, 1 << 20)] + [[float("nan")] * 1000 for _ in range(50)]
                del junk
            if args.dev_churn:
                junk = [torch.full((n,), float("nan"), device="cuda") for n in (16, 256, 4096, 65536, 1 << 20, 1 << 24, 1 << 27)]
                del junk
            if args.events:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                evs.append(ev)
            if args.pre and i % 4 == 3 and not args.no_sync:
                print("   timing step %d: %s" % (i, m.get_current_errors()), flush=True)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("   after the timing loop: %s" % dict(m.get_current_errors()), flush=True)
        out[mode] = (losses, w)
        print("%-5s  %.3f ms/step (host %.3f ms/step)  %.1f images/s   recordings %s" %
              (mode, t_all * 1e3 / args.time_steps, t_host * 1e3 / args.time_steps, args.batch * args.time_steps / t_all, sg.recordings if sg else "-"), flush=True)
        for i, l in enumerate(losses):
            print("   step %d  G_GAN %.6f  G_L1 %.6f  D %.6f  F %.6f" % (i, *l), flush=True)
        del m, sg
        torch.cuda.empty_cache()
    for a, b in (("eager", "warm"), ("warm", "graph")):
        if a not in out or b not in out:
            continue
        la, wa = out[a]; lb, wb = out[b]
        print("%s vs %s: losses identical %s;  weights: bitwise equal %s, max |diff| %.3e, max |w| %.3e" %
              (a, b, la == lb, torch.equal(wa, wb), (wa - wb).abs().max().item(), wa.abs().max().item()))


if __name__ == "__main__":
    main()
