#!/usr/bin/env python3
"""bench.py — training throughput of the IPSR inpainting step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF: the parent — before any GPU call, and
never touching the GPU afterwards — runs the second command above as a CHILD process (no exec), lets rank 0's JSON line through and
exits with the child's status.  `--backend gloo` is a dry mode for machines without GPUs (launcher, argument plumbing, the bucketed
gradient exchange and the N>1 reporting on small stand-in nets; no kernels, labelled `dry_run`).

One "step" = the reference's training iteration (train.ipynb cell 2:24-27) on device-resident synthetic
tensors of BASELINE.json config 2:  set_input + set_ref_latent + set_gt_latent + optimize_parameters,
256x256, 128x128 centre hole (M = 256 masked feature positions), batch 8 per GPU, fp32, dropout on
(train.ipynb default), random-init nets and a seeded random VGG16 (no network access).  Nothing is
skipped inside the timed region: 4 nets forward/backward, 3 VGG passes (the reference's 4th is a
recomputation of identical features and is reused), 4 Adam steps, the IPSR layer forward/backward.

Rank 0 prints ONE JSON line.  `value` = images/s of the whole job (all ranks).  Extra objects:
  roofline      the layer's dominant kernel (fp32-MFMA correlation + arg-max): algorithmic FLOPs per launch
                (2*N*N*C per sample) over its mean duration, measured live with HIP events on
                the launch stream inside the timed steps (C-ABI hook ipsr_profile_*).
  conv_roofline the convolutions' matrix-core kernel (ipsr::wino_gemm_kernel, all its launches in the timed steps): executed
                flops over summed launch time, HIP events on the launch stream (region 3 of the C-ABI hook).
  cpu_baseline  the CPU twin (oracle C restatement of the layer + PyTorch-CPU convs, oracle/cpu_model.py)
                timed on this host's cores on a bounded sample (rank 0, N=1 only).
  ipsr_layer_ms IPSR layer forward+backward at the same shape (second half of BASELINE.json's metric): `in_step` = HIP
                events around every ipsr_forward / ipsr_backward call INSIDE the timed steps (the real, signed conv
                features: truncation survivors make the backward slower than on synthetic features), `standalone` = the
                layer alone on synthetic non-negative features (median of 50).
  strict_reference  the same step with opt.strict_reference=True (the reference's exact sequence incl. the work whose
                results it never reads, 414.9 GFLOP/image), timed after the headline loop.
  step_graph    whether the timed steps were replayed from one HIP graph (deepinpainting_amd/stepgraph.py; --graph on, off by
                default: the replay is no faster than the eager step here); the per-kernel event timings are then taken in eager
                steps right after the loop.  `host_enqueue_ms_per_step` = the host's share of a step (kernels queued, nothing waited for).
`value` = batch * world * K / wall time of the K timed steps (the driver's contract); `ms_per_step_median` and
`images_per_sec_median_step` (SURVEY §8d's definition) come from HIP events recorded at every step boundary.
"""
import argparse
import contextlib
import ctypes
import io
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


from deepinpainting_amd import use_shipped_miopen_db  # noqa: E402

use_shipped_miopen_db()      # a private per-rank copy of the shipped MIOpen find-db; must precede the first convolution

STEP_FLOPS_PER_IMAGE = 340.6e9     # conv/mm forward+backward of one training step at 256x256 AS EXECUTED here: torch
                                   # FlopCounterMode over the default trainer = 339.5, + 1.07 of the IPSR correlation.  The
                                   # reference's sequence is 414.9; the difference is work whose results it never
                                   # uses (duplicate VGG pass, VGG slice 4 of the generated image, the discriminator gradients
                                   # of backward_G, the layer's two dense N x N GEMMs) — tests/test_host_model.py::
                                   # test_default_mode_changes_no_live_value
METRIC = "train images/sec at 256x256, batch 8/GPU, 1/2/4/8 MI355X; IPSR layer ms"
FINE, BATCH, C_FEAT, H_FEAT = 256, 8, 512, 32
REF_FLOPS_PER_IMAGE = 414.9e9      # the reference's own sequence (torch FlopCounterMode)
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32-input MFMA = 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA, same guide (the 5 PF headline figure includes 2:1 sparsity)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_corr_argmax.json")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def synthetic_batch(device, batch, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    img = torch.rand(batch, 3, FINE, FINE, device=device, generator=g) * 2 - 1
    ref = torch.rand(batch, 3, FINE, FINE, device=device, generator=g) * 2 - 1
    mask = torch.zeros(1, 1, FINE, FINE, dtype=torch.bool, device=device)
    mask[:, :, FINE // 4:3 * FINE // 4, FINE // 4:3 * FINE // 4] = 1
    return img, mask, ref


def train_step(model, img, mask, ref):
    model.set_input(img, mask, ref)
    model.set_ref_latent()
    model.set_gt_latent()
    model.optimize_parameters()


def _layer_case(device, B, h, fine, mask_img, patch, iters, corr="fp32"):
    """Median forward / backward ms of the IPSR layer alone on synthetic features (`x = |N(0,1)|`, `ref = relu(N(0,1))`)."""
    from deepinpainting_amd import ops
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(B, C_FEAT, h, h, device=device, generator=g).abs()
    ref = torch.relu(torch.randn(B, C_FEAT, h, h, device=device, generator=g))
    grad = torch.randn(B, C_FEAT, h, h, device=device, generator=g)
    feat = ops.feat_mask(mask_img, 3, 5 / 16.0)
    _, mpi, cnt = ops.index_prep(feat, patch, 1, 1)
    M = int(cnt.item())
    mpi = mpi[:M].contiguous()
    for _ in range(3):
        f = ops.forward(x, ref, mpi, patch=patch, corr=corr)
        ops.backward(grad, f.bwd_index, 1.0, M, patch=patch)
    torch.cuda.synchronize()
    fwd, bwd = [], []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(iters):
        ev[0].record()
        f = ops.forward(x, ref, mpi, patch=patch, corr=corr)
        ev[1].record()
        ops.backward(grad, f.bwd_index, 1.0, M, patch=patch)
        ev[2].record()
        torch.cuda.synchronize()
        fwd.append(ev[0].elapsed_time(ev[1]))
        bwd.append(ev[1].elapsed_time(ev[2]))
    return statistics.median(fwd), statistics.median(bwd), M


def new_mask_each_step(model, img, ref, device, batch, ksteps):
    """BASELINE config 3's real case (train.ipynb c2:16-19 draws a fresh free-form mask per iteration): every step gets a NEW mask tensor,
    so K1 (3 box filters) + K2 (index prep) and the one host read of the masked count (models/IPSR_model.py: capacity of the device-side
    index) sit inside the timed region — the headline loop reuses one mask tensor, for which set_mask returns early."""
    from deepinpainting_amd.util.staging import random_stroke_mask
    masks = [random_stroke_mask(FINE, torch.Generator().manual_seed(500 + i), device=device) for i in range(ksteps + 2)]
    if os.environ.get("IPSR_BENCH_INDEX_CAP"):                # A/B knob: "full" = never a host read, the layer's buffers sized by N
        model.CSA_model[0].index_capacity = os.environ["IPSR_BENCH_INDEX_CAP"]
    for i in range(2):
        train_step(model, img, masks[i], ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(ksteps):
        train_step(model, img, masks[2 + i], ref)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(batch * ksteps / dt, 3), "unit": "images/sec", "ms_per_step": round(dt / ksteps * 1e3, 3), "steps": ksteps,
            "masked_feature_positions": [int(model.CSA_model[0].mask_point_idx.numel())],
            "note": "a fresh free-form (random stroke) mask tensor per step: feature-mask pyramid, index prep and the host read of the masked count inside the timed region"}


def _centre(device, size):
    m = torch.zeros(size, size, dtype=torch.uint8, device=device)
    m[size // 4:3 * size // 4, size // 4:3 * size // 4] = 1
    return m


def layer_timing(device, iters=50):
    """IPSR layer forward+backward (ms, median) at [8,512,32,32], M=256 (BASELINE config 2)."""
    f, b, _ = _layer_case(device, BATCH, H_FEAT, FINE, _centre(device, FINE), 1, iters)
    return f, b


def layer_timing_bf16corr(device, iters=50):
    """BASELINE config 5's layer: the same [8,512,32,32] / M=256 case with the correlation on the bf16 MFMA kernel, the
    kernel's own time (HIP events through the C-ABI hook) and its arg-max agreement with the fp32 kernel."""
    from deepinpainting_amd import _lib, ops
    lib = _lib.lib()
    f, b, _ = _layer_case(device, BATCH, H_FEAT, FINE, _centre(device, FINE), 1, iters, corr="bf16")
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(BATCH, C_FEAT, H_FEAT * H_FEAT, device=device, generator=g).abs()
    ref = torch.relu(torch.randn(BATCH, C_FEAT, H_FEAT * H_FEAT, device=device, generator=g))
    xn, _ = ops.patch_normalize(x)
    i32, _, _ = ops.corr_argmax(xn, ref)
    lib.ipsr_profile_enable(20)
    for _ in range(20):
        i16, _, _ = ops.corr_argmax(xn, ref, corr="bf16")
    buf = (ctypes.c_float * 20)()
    n = lib.ipsr_profile_read_region(0, ctypes.cast(buf, ctypes.c_void_p), 20)
    lib.ipsr_profile_enable(0)
    kms = statistics.median([buf[i] for i in range(n)]) if n else None
    flops = 2.0 * (H_FEAT * H_FEAT) ** 2 * C_FEAT * BATCH
    return {"dtype": "bf16", "forward": round(f, 4), "backward": round(b, 4), "total": round(f + b, 4),
            "shape": "[%d,%d,%d,%d], M=256, correlation on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16, fp32 accumulate), "
                     "rest of the layer fp32" % (BATCH, C_FEAT, H_FEAT, H_FEAT),
            "corr_kernel_ms": round(kms, 5) if kms else None,
            "corr_kernel_tflops": round(flops / (kms * 1e-3) / 1e12, 1) if kms else None,
            "argmax_agreement_with_fp32": round(float((i16 == i32).float().mean().item()), 5)}


def layer_timing_other_configs(device):
    """The layer at the other BASELINE.json configurations (parity-test cases, reported for reference only)."""
    from deepinpainting_amd.util.staging import random_stroke_mask
    out = {}
    stroke = random_stroke_mask(FINE, torch.Generator().manual_seed(3), device=device)[0, 0].to(torch.uint8)
    for key, (B, h, fine, mimg, patch, iters) in {
            "config3_freeform_256": (BATCH, H_FEAT, FINE, stroke, 1, 20),
            "config4_512_patch1": (4, 64, 512, _centre(device, 512), 1, 10),
            "config4_512_patch3": (4, 64, 512, _centre(device, 512), 3, 5)}.items():
        f, b, M = _layer_case(device, B, h, fine, mimg, patch, iters)
        out[key] = {"forward": round(f, 4), "backward": round(b, 4), "shape": "[%d,%d,%d,%d], shift_sz=%d, M=%d" % (B, C_FEAT, h, h, patch, M)}
    out["config3_per_sample_masks_256"] = _per_sample_case(device)
    return out


def _per_sample_case(device, iters=10):
    """Extension: one free-form mask PER SAMPLE (models/IPSR_model.py) — the layer then runs sample by sample."""
    from collections import namedtuple
    from deepinpainting_amd.models.IPSR_model import IPSR_model
    from deepinpainting_amd.util.staging import random_stroke_mask
    Vgg = namedtuple("VggOutputs", ["relu1_2", "relu2_2", "relu3_3", "relu4_3"])
    g = torch.Generator(device=device).manual_seed(7)
    x = torch.randn(BATCH, C_FEAT, H_FEAT, H_FEAT, device=device, generator=g).abs().requires_grad_(True)
    ref = Vgg(None, None, None, torch.relu(torch.randn(BATCH, C_FEAT, H_FEAT, H_FEAT, device=device, generator=g)))
    grad = torch.randn(BATCH, C_FEAT, H_FEAT, H_FEAT, device=device, generator=g)
    masks = torch.cat([random_stroke_mask(FINE, torch.Generator().manual_seed(100 + b), device=device) for b in range(BATCH)], 0)
    layer = IPSR_model(5 / 16.0, 1, 1, 1, 1, 1.0)
    layer.set_mask(masks, 3, 5 / 16.0)
    layer.set_ref(ref)
    fwd, bwd = [], []
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for i in range(iters + 2):
        xin = x * 1.0                     # a non-leaf, as in the network (a leaf's AccumulateGrad lives on ONE stream)
        ev[0].record()
        y = layer(xin)
        ev[1].record()
        torch.autograd.grad(y, xin, grad)
        ev[2].record()
        torch.cuda.synchronize()
        if i >= 2:
            fwd.append(ev[0].elapsed_time(ev[1]))
            bwd.append(ev[1].elapsed_time(ev[2]))
    Ms = [int(ix[3].numel()) for ix in layer._per_sample_index]
    return {"forward": round(statistics.median(fwd), 4), "backward": round(statistics.median(bwd), 4),
            "shape": "[%d,%d,%d,%d], M per sample %d..%d (incl. the autograd wrapper)" % (BATCH, C_FEAT, H_FEAT, H_FEAT, min(Ms), max(Ms))}


def cpu_baseline(sample_batch=BATCH, steps=1):
    """The CPU twin on this host's cores: same trainer class, PyTorch-CPU convs, oracle-backed IPSR layer."""
    from deepinpainting_amd.options import Option
    from oracle import cpu_model, ipsr_oracle as orc
    orc.build()
    cores = torch.get_num_threads()
    opt = Option(gpu_ids=[], batchSize=sample_batch, use_dropout=True, quiet=True, allow_random_vgg=True,
                 checkpoints_dir=os.path.join("/tmp", "ipsr_bench_ckpt_cpu"))
    torch.manual_seed(1234)
    model = quiet(cpu_model.create_cpu_model, opt)
    img, mask, ref = synthetic_batch(torch.device("cpu"), sample_batch, 1234)
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, img, mask, ref)
    dt = time.perf_counter() - t0
    # layer alone (oracle C restatement) at the SAME shape as the GPU figure: [8,512,32,32], M=256, forward + backward
    import numpy as np
    rs = np.random.RandomState(7)
    x = np.abs(rs.standard_normal((BATCH, C_FEAT, H_FEAT, H_FEAT))).astype(np.float32)
    rf = np.maximum(rs.standard_normal((BATCH, C_FEAT, H_FEAT, H_FEAT)), 0).astype(np.float32)
    mpi = model.CSA_model[0].mask_point_idx
    layer = []
    for i in range(4):                          # one warm-up call (OpenMP thread start-up, page faults), then the median of three
        t1 = time.perf_counter()
        f = orc.forward(x, rf, mpi)
        orc.backward(x, mpi, f.attn_rows, f.bwd_index, 1.0)
        if i:
            layer.append((time.perf_counter() - t1) * 1e3)
    layer_ms = statistics.median(layer)
    return {
        "value": round(sample_batch * steps / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
        "sample": "%d full training step(s) at batch %d — the GPU figure's own batch — (same 256x256 workload, oracle-backed IPSR layer + "
                  "PyTorch-CPU convs, torch %s, %d threads): %.1f s" % (steps, sample_batch, torch.__version__, cores, dt),
        "ipsr_layer_ms": round(layer_ms, 2),
        "ipsr_layer_shape": "[%d,%d,%d,%d], M=%d, forward+backward, oracle C restatement (OpenMP over samples / column blocks), warm median of 3" % (BATCH, C_FEAT, H_FEAT, H_FEAT, len(mpi)),
    }


T_START = time.perf_counter()


def conv_roofline(args, ng, gsteps, gemm_ms, gemm_flops, gemm_useful):
    """The Winograd GEMM launches of the timed steps against the matrix-core peak of the arithmetic they ran in: fp32 MFMA (157 TF) for
    the fp32 nets, the dense bf16 MFMA peak for the split-bf16 kernels — there every logical multiply is 3 (bf16x3) or 6 (bf16x6)
    bf16 MFMA products, so the hardware fraction counts them (`mfma_products_per_multiply`)."""
    split = args.dtype != "f32" or args.conv_math != "fp32"
    math = args.conv_math if args.dtype == "f32" else "bf16x3"          # opt.conv_math_bf16's default
    k = {"bf16x3": 3, "bf16x6": 6}.get(math, 1) if split else 1
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    sec = gemm_ms * 1e-3
    return {"kernel": ("ipsr::wino_gemm_split_kernel (bf16 MFMA on split operands, %s)" % math) if split
            else "ipsr::wino_gemm_kernel (fp32 MFMA, 36 GEMMs per convolution)", "bound": "mfma",
            "achieved": round(gemm_useful / sec / 1e12, 2) if ng else None, "peak": peak, "unit": "TFLOP/s",
            "mfma_products_per_multiply": k,
            "frac": round(k * gemm_useful / sec / 1e12 / peak, 4) if ng else None,
            "executed_tflops": round(gemm_flops / sec / 1e12, 2) if ng else None,
            "executed_frac": round(k * gemm_flops / sec / 1e12 / peak, 4) if ng else None,
            "padding_share_of_executed_flops": round(1.0 - gemm_useful / gemm_flops, 4) if ng and gemm_flops else None,
            "launches_timed": ng, "launches_per_step": round(ng / gsteps, 1), "kernel_ms_per_step": round(gemm_ms / gsteps, 3),
            "flop_saving_vs_direct": "F(4x4,3x3) and F(3x3,4x4): 4.0x; polyphase F(5x5,2x2): 2.78x (per family, not applied here)"}


def self_launch(n, argv):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as a CHILD process and return its exit status.
    Runs before this process has made any GPU call (importing torch makes none) and makes none afterwards: on this pool a process
    that initialised the GPU must never exec, and N ranks + an idle parent holding the card would only cost memory."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # this pool's driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    print("[bench] --gpus %d without a torchrun environment: starting %s" % (n, " ".join(cmd[1:])), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env)                    # stdout / stderr inherited: rank 0's JSON line goes straight through
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def rccl_env():
    return {k: v for k, v in sorted(os.environ.items())
            if k.startswith(("NCCL_", "RCCL_")) or k in ("HSA_ENABLE_IPC_MODE_LEGACY", "HSA_FORCE_FINE_GRAIN_PCIE", "OMP_NUM_THREADS")}


def ddp_report(args, world, rank, device, elapsed, step_ms, reducers, backend):
    """N > 1 only (collective: every rank calls it).  What the exchange looked like from every rank: the ranks the backend saw, each
    rank's own time per step, and the EXPOSED part of the two gradient all-reduces — the time the compute stream waits in
    GradBucketReducer.finish() after the backward's last kernel (HIP events on that stream; the rest ran under the backward)."""
    import torch.distributed as dist
    mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(),
            "device": (torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu"),
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
            "ms_per_step_median": round(statistics.median(step_ms), 3) if step_ms else None}
    for key, red in reducers.items():
        ms = red.exposed_ms() if red is not None else []
        ms = ms[-args.steps:] if args.steps else ms
        mine["exposed_allreduce_ms_" + key] = round(statistics.mean(ms), 4) if ms else None
    if device.type == "cuda":
        mine["hbm_peak_allocated_gb"] = round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank != 0:
        return None
    exp = {k: [g.get("exposed_allreduce_ms_" + k) for g in gathered] for k in reducers}
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else " (CPU dry mode)"), "world_size_seen": dist.get_world_size(),
            "ranks": gathered,
            "exposed_allreduce_ms_per_step": {k: {"max_over_ranks": max(v) if all(x is not None for x in v) else None,
                                                  "mean_over_ranks": round(statistics.mean(v), 4) if all(x is not None for x in v) else None}
                                              for k, v in exp.items()},
            "allreduce_bytes_per_step": {k: (red.bytes_per_exchange() if red is not None else None) for k, red in reducers.items()},
            "buckets": {k: (len(red.buckets) if red is not None else None) for k, red in reducers.items()},
            "bucket_mb": args.ddp_bucket_mb, "env": rccl_env()}


def dry_run(args):
    """`--backend gloo`: everything of the N>1 path that is not a kernel — rendezvous from the torchrun environment, rank-0 broadcast,
    two bucketed gradient exchanges per step ((D,F) then (G,P), as models/IPSR.py orders them) on small stand-in nets, the barrier /
    max-over-ranks timing and the N>1 report — on the CPU.  The line says `dry_run`; its `value` is not a measurement of anything."""
    import torch.nn as nn
    from deepinpainting_amd import dist as idist
    rank, world, _ = idist.init_distributed(backend="gloo")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    device = torch.device("cpu")
    torch.manual_seed(1234 + rank)                   # different init per rank: the broadcast has to make them equal

    def net(cin, cout):
        return nn.Sequential(nn.Conv2d(cin, 16, 3, padding=1), nn.InstanceNorm2d(16, affine=True), nn.ReLU(), nn.Conv2d(16, cout, 3, padding=1))
    nets = {"G": net(6, 3), "P": net(3, 3), "D": net(3, 1), "F": net(3, 1)}
    for m in nets.values():
        idist.broadcast_module(m, src=0)
    reducers = {"D": idist.GradBucketReducer([nets["D"], nets["F"]], bucket_bytes=2048),
                "G": idist.GradBucketReducer([nets["G"], nets["P"]], bucket_bytes=2048)}
    for r in reducers.values():
        r.timing = True
    opts = {k: torch.optim.SGD(m.parameters(), lr=1e-2) for k, m in nets.items()}
    g = torch.Generator().manual_seed(99 + rank)
    img = torch.rand(args.batch, 3, 16, 16, generator=g)

    def step():
        fake_p = nets["P"](img)
        fake = nets["G"](torch.cat((fake_p, img), 1))
        for k in ("D", "F"):
            opts[k].zero_grad()
        reducers["D"].arm()
        (nets["D"](fake.detach()).pow(2).mean() + nets["F"](img).pow(2).mean()).backward()
        reducers["D"].finish()
        for k in ("D", "F"):
            opts[k].step()
        for k in ("G", "P"):
            opts[k].zero_grad()
        frozen = [p for k in ("D", "F") for p in nets[k].parameters()]
        for p in frozen:
            p.requires_grad_(False)
        reducers["G"].arm()
        ((fake - img).abs().mean() + nets["D"](fake).pow(2).mean()).backward()
        reducers["G"].finish()
        for p in frozen:
            p.requires_grad_(True)
        for k in ("G", "P"):
            opts[k].step()
    for _ in range(args.warmup):
        step()
    for r in reducers.values():
        r.exposed_ms()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    flat = torch.cat([p.detach().reshape(-1) for m in nets.values() for p in m.parameters()])
    ddp = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ddp = ddp_report(args, world, rank, device, elapsed, step_ms, reducers, "gloo")
        elapsed = float(t.item())
        lo, hi = flat.clone(), flat.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))
    else:
        in_sync = True
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(args.batch * world * args.steps / elapsed, 3), "unit": "images/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "dry_run": True,
                          "config": {"workload": "DRY RUN (--backend gloo): stand-in nets on the CPU, no HIP kernels — launcher, rendezvous, gradient "
                                                 "exchange and reporting of the N>1 path only; not a measurement",
                                     "global_batch": args.batch * world, "parallelism": "dp%d" % world},
                          "weights_identical_on_all_ranks_after_the_steps": in_sync, "ddp": ddp}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if in_sync else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (BASELINE config 2 = 8)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 = the reference's precision and the headline metric; bf16 = BASELINE config 5 (convs under "
                         "bf16 autocast, IPSR layer and losses fp32) — a separate, clearly labelled measurement")
    ap.add_argument("--rccl-algo", default=None, help="NCCL_ALGO for the gradient all-reduce (default: RCCL's tuner)")
    ap.add_argument("--rccl-proto", default=None, help="NCCL_PROTO (e.g. Simple)")
    ap.add_argument("--rccl-min-channels", default=None, help="NCCL_MIN_NCHANNELS")
    ap.add_argument("--conv-math", choices=("fp32", "bf16x6", "bf16x3"), default="fp32",
                    help="arithmetic of the Winograd convolution GEMMs for fp32 activations (default fp32 = the reference's; the split-bf16 "
                         "forms are reported as what they are, never as the headline)")
    ap.add_argument("--debug-opt", action="append", default=[], help="key=value for ipsr_debug_set_option (kernel-variant A/B; never set by default)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL over xGMI (the measurement); gloo = DRY MODE on the CPU: launcher + argument plumbing + the "
                         "gradient exchange on stand-in nets, no kernels (tests/test_bench_launcher.py)")
    ap.add_argument("--ddp-bucket-mb", type=int, default=64, help="gradient bucket size of the all-reduce (dist.GradBucketReducer)")
    ap.add_argument("--graph", choices=("on", "off"), default="off",
                    help="on: replay the training step from one HIP graph (deepinpainting_amd/stepgraph.py, one GPU only) instead of queueing its "
                         "~1100-1400 kernels from Python every step.  Off by default: measured at --dtype bf16 --batch 16 the replay takes 22.0 ms "
                         "against 21.2 ms for the eager step (hipGraphLaunch orders its ~1400 nodes one by one), although it frees the host "
                         "(0.4 ms instead of 21.9 ms of queueing per step)")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.backend == "gloo":
        return dry_run(args)
    if args.debug_opt:
        from deepinpainting_amd import _lib as _dbg_lib
        for kv in args.debug_opt:
            k, v = map(int, kv.split("="))
            _dbg_lib.check(_dbg_lib.lib().ipsr_debug_set_option(k, v), "ipsr_debug_set_option")

    from deepinpainting_amd import _lib, dist as idist
    from deepinpainting_amd.models.models import create_model
    from deepinpainting_amd.options import Option

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the IPSR layer has no CPU path (the CPU twin is only the reported baseline)")
    # the device is chosen from the launcher's LOCAL_RANK before the communicator exists; init_distributed sets it again for nccl
    rank, world, local_rank = idist.init_distributed(backend="nccl", rccl_algo=args.rccl_algo, rccl_proto=args.rccl_proto,
                                                        rccl_channels=args.rccl_min_channels)
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MIOpen exhaustive find (cudnn.benchmark) costs minutes on a fresh box for the ~60 conv shapes of the
    # step; off by default so the run finishes quickly, opt in with IPSR_BENCH_MIOPEN_FIND=1
    torch.backends.cudnn.benchmark = os.environ.get("IPSR_BENCH_MIOPEN_FIND", "0") == "1"

    opt = Option(gpu_ids=[local_rank], batchSize=args.batch, use_dropout=True, quiet=True, allow_random_vgg=True,
                 batch_vgg=os.environ.get("IPSR_BENCH_BATCH_VGG", "0") == "1", batch_disc=os.environ.get("IPSR_BENCH_BATCH_DISC", "1") == "1", amp_bf16=(args.dtype == "bf16"), conv_math=args.conv_math,
                 ddp_bucket_mb=args.ddp_bucket_mb, checkpoints_dir=os.path.join("/tmp", "ipsr_bench_ckpt_%d" % rank))
    torch.manual_seed(1234)                       # identical init on every rank (rank 0 is broadcast anyway)
    model = quiet(create_model, opt)
    reducers = {"D": model._reducer_D, "G": model._reducer_G}
    for red in reducers.values():
        if red is not None:
            red.timing = world > 1
    cl = os.environ.get("IPSR_BENCH_CHANNELS_LAST", "0")             # experiment knob, not the default: "1" = all nets,
    if cl != "0":                                                    # or a comma list of netG,netP,netD,netF,vgg
        names = ("netG", "netP", "netD", "netF", "vgg") if cl == "1" else tuple(cl.split(","))
        for name in names:
            getattr(model, name).to(memory_format=torch.channels_last)
    img, mask, ref = synthetic_batch(device, args.batch, 1234 + rank)

    lib = _lib.lib()
    for i in range(args.warmup):
        train_step(model, img, mask, ref)
        torch.cuda.synchronize()
        if rank == 0:
            print("[bench] warm-up step %d/%d done (%.1f s since start)" % (i + 1, args.warmup, time.perf_counter() - T_START),
                  file=sys.stderr, flush=True)
    # A CPython gen-2 collection walks every live object (the module trees, ~100 ms here) and would land inside one
    # timed step (measured: one 123 ms step in every ~25).  Collect now and freeze the survivors out of future scans.
    import gc
    gc.collect()
    gc.freeze()
    for red in reducers.values():
        if red is not None:
            red.exposed_ms()                      # drop the warm-up steps' records
    # One GPU: the step is replayed from a HIP graph recorded here (the recording's own warm-up steps are undone, see StepGraph._record);
    # if the recording fails the loop below queues the kernels from Python as before, and the JSON line says so.
    graph_info = {"used": False, "reason": "--graph off (the default)" if args.graph == "off" else "N > 1: the gradient exchange is not recorded"}
    sgraph = None
    if world == 1 and args.graph == "on":
        from deepinpainting_amd.stepgraph import StepGraph
        try:
            sgraph = StepGraph(model)
            sgraph.step(img, mask, ref)
            torch.cuda.synchronize()
            graph_info = {"used": True, "recordings": sgraph.recordings,
                          "note": "timed steps = hipGraphLaunch of the recorded step (same kernels, same order); the per-kernel HIP-event "
                                  "timings of this line are taken in eager steps right after the timed loop (events are not recorded into a graph)"}
        except Exception as e:                                      # noqa: BLE001 — report and measure the eager step instead
            sgraph = None
            graph_info = {"used": False, "reason": "recording failed: %s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.synchronize()
        if rank == 0:
            print("[bench] step graph: %s (%.1f s since start)" % (graph_info, time.perf_counter() - T_START), file=sys.stderr, flush=True)

    def one_step():
        if sgraph is not None:
            sgraph.step(img, mask, ref)
        else:
            train_step(model, img, mask, ref)
    if sgraph is None:
        lib.ipsr_profile_enable(max(args.steps, 1))
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        one_step()
        step_ev[i + 1].record()
    host_enqueue_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)      # the host's share: launches queued, nothing waited for
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    # a step that went wrong must not pass for a throughput number: the losses of the last timed step are reported and flagged
    timed_losses = {k: float(v) for k, v in model.get_current_errors().items()}
    losses_finite = all(v == v and abs(v) != float("inf") for v in timed_losses.values())
    if not losses_finite and rank == 0:
        print("[bench] WARNING: non-finite losses after the timed steps: %s — the rate below measures a broken step" % timed_losses, file=sys.stderr, flush=True)
    ddp = ddp_report(args, world, rank, device, elapsed, step_ms, reducers, "nccl") if world > 1 else None
    for red in reducers.values():
        if red is not None:
            red.timing = False

    def read_region(region):
        buf = (ctypes.c_float * max(args.steps, 1))()
        n = lib.ipsr_profile_read_region(region, ctypes.cast(buf, ctypes.c_void_p), args.steps)
        return [buf[i] for i in range(n)]
    if sgraph is None:
        kern_ms, fwd_in_step, bwd_in_step = read_region(0), read_region(1), read_region(2)
    lib.ipsr_profile_enable(0)
    # region 3: every launch of the convolutions' matrix-core kernel (Winograd GEMM), with its flops — in a few EXTRA steps
    # after the timed loop (~100 more event records per step would perturb the headline number)
    gsteps = 3
    ncap = 256 * gsteps
    lib.ipsr_profile_enable_mask(gsteps, 0x38 if sgraph is None else 0x3F)
    for _ in range(gsteps):
        train_step(model, img, mask, ref)
    torch.cuda.synchronize()
    if sgraph is not None:                                # the layer's regions too: the timed loop replayed a graph, which carries no events
        kern_ms, fwd_in_step, bwd_in_step = read_region(0), read_region(1), read_region(2)

    def read_work(region):
        ms, work, use = (ctypes.c_float * ncap)(), (ctypes.c_double * ncap)(), (ctypes.c_double * ncap)()
        n = lib.ipsr_profile_read_region_work2(region, ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(work, ctypes.c_void_p), ctypes.cast(use, ctypes.c_void_p), ncap)
        return n, ms, work, use
    ng, gms, gwork, guse = read_work(3)
    nd, dms, dwork, duse = read_work(4)                   # the direct bf16 convolution kernels (config 5 only)
    ni, ims, iwork, _ = read_work(5)                      # the InnerCos / InnerCos2 loss taps inside the training steps
    ic_step = None
    if ni:
        ic_ms = statistics.median(ims[i] for i in range(ni))
        ic_step = {"kernel": "ipsr::innercos_fused_kernel (masked MSE tap: the streaming kernel; a 256-thread kernel then folds its block partials)", "bound": "hbm",
                   "achieved": round(iwork[0] / (ic_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(iwork[0] / (ic_ms * 1e-3) / 1e9 / 8000.0, 4),
                   "bytes_per_launch": iwork[0], "kernel_ms": round(ic_ms, 5), "launches_timed": ni,
                   "note": "HIP events on the launch stream around every launch inside %d training steps (median); algorithmic bytes = x + target read once" % gsteps}
    gemm_ms, gemm_flops = sum(gms[i] for i in range(ng)), sum(gwork[i] for i in range(ng))
    gemm_useful = sum(guse[i] for i in range(ng))
    direct = None
    if nd:
        d_ms, d_flops, d_use = sum(dms[i] for i in range(nd)), sum(dwork[i] for i in range(nd)), sum(duse[i] for i in range(nd))
        direct = {"kernel": "ipsr::conv_bf16_kernel / conv_bf16_wrw_kernel (direct implicit GEMM on v_mfma_f32_32x32x16_bf16, bf16 operands, fp32 accumulate)",
                  "bound": "mfma", "achieved": round(d_use / (d_ms * 1e-3) / 1e12, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(d_use / (d_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4), "executed_tflops": round(d_flops / (d_ms * 1e-3) / 1e12, 2),
                  "padding_share_of_executed_flops": round(1.0 - d_use / d_flops, 4), "launches_per_step": round(nd / gsteps, 1),
                  "kernel_ms_per_step": round(d_ms / gsteps, 3)}
    if os.environ.get("IPSR_BENCH_GEMM_DUMP"):          # one line per launch: ms, flops, TFLOP/s (for tuning the tile/split choice)
        with open(os.environ["IPSR_BENCH_GEMM_DUMP"], "w") as fh:
            for i in range(ng):
                fh.write("%d %.4f %.0f %.1f\n" % (i, gms[i], gwork[i], gwork[i] / max(gms[i], 1e-6) / 1e9))
    lib.ipsr_profile_enable(0)

    # the reference's exact sequence (strict_reference), same model and inputs, timed the same way after the headline loop
    strict = None
    if args.dtype == "f32" and os.environ.get("IPSR_BENCH_NO_STRICT", "0") != "1" and os.environ.get("IPSR_BENCH_STEP_ONLY", "0") != "1":
        ksteps = max(3, min(10, args.steps))
        model.strict_reference = True
        for _ in range(2):
            train_step(model, img, mask, ref)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        ts = time.perf_counter()
        for _ in range(ksteps):
            train_step(model, img, mask, ref)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        strict_elapsed = time.perf_counter() - ts
        model.strict_reference = False
        if world > 1:
            t = torch.tensor([strict_elapsed], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            strict_elapsed = float(t.item())
        strict = {"value": round(args.batch * world * ksteps / strict_elapsed, 3), "unit": "images/sec", "steps": ksteps,
                  "ms_per_step": round(strict_elapsed / ksteps * 1e3, 3), "flops_per_image": REF_FLOPS_PER_IMAGE,
                  "note": "opt.strict_reference=True: the reference's sequence incl. the duplicate VGG pass, VGG slice 4 of "
                          "the generated image and the discriminator gradients of backward_G (dead work, bit-identical live values)"}

    # the same step with the Winograd GEMMs on split-bf16 operands (bf16 MFMA, fp32 accumulate) — separately labelled, NEVER the
    # headline: bf16x3 = hi + lo planes (error ~1e-4 of a convolution's output scale), bf16x6 = three planes (the fp32 path's accuracy)
    alt = None
    if args.dtype == "f32" and args.conv_math == "fp32" and world == 1 and os.environ.get("IPSR_BENCH_NO_ALT", "0") != "1" \
            and os.environ.get("IPSR_BENCH_STEP_ONLY", "0") != "1":
        from deepinpainting_amd.models import hipconv
        alt = {}
        for math in ("bf16x3", "bf16x6"):
            hipconv.set_conv_math(fp32=math)
            for _ in range(2):
                train_step(model, img, mask, ref)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            ksteps = max(3, min(10, args.steps))
            for _ in range(ksteps):
                train_step(model, img, mask, ref)
            torch.cuda.synchronize()
            dt = time.perf_counter() - ts
            alt[math] = {"value": round(args.batch * ksteps / dt, 3), "unit": "images/sec", "ms_per_step": round(dt / ksteps * 1e3, 3), "steps": ksteps}
        hipconv.set_conv_math(fp32="fp32")
        alt["note"] = ("opt.conv_math: transformed Winograd operands split into 2 / 3 bf16 numbers, multiplied on v_mfma_f32_32x32x16_bf16 with fp32 "
                       "accumulation; measured convolution error vs fp64: fp32 path 1.4e-5, bf16x6 1.4e-5, bf16x3 1.4e-4 of the output scale "
                       "(tests/test_gpu_conv.py::test_split_bf16_winograd_arithmetic_all_families).  Not the headline arithmetic.")

    newmask = None
    if args.dtype == "f32" and world == 1 and os.environ.get("IPSR_BENCH_STEP_ONLY", "0") != "1":
        newmask = new_mask_each_step(model, img, ref, device, args.batch, max(3, min(10, args.steps)))

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    errs = model.get_current_errors()
    ms_per_step = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    n_feat = H_FEAT * H_FEAT
    flops = 2.0 * n_feat * n_feat * C_FEAT * args.batch
    corr_peak = PEAK_FP32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS      # the kernel of the layer that actually ran
    kms = statistics.mean(kern_ms) if kern_ms else float("nan")
    achieved = flops / (kms * 1e-3) / 1e12 if kern_ms else None
    traffic, traffic_src = None, None
    if os.path.exists(TRAFFIC_FILE):
        with open(TRAFFIC_FILE) as fh:
            tj = json.load(fh)
        # a stored PMC figure (two separate rocprofv3 --pmc passes, tools/collect_traffic.py), NOT re-measured in this run;
        # only quoted when it was taken on this kernel at this shape
        if "corr_argmax" in tj.get("kernel", "") and tj.get("workload", "").endswith("[%d,%d,%d,%d]" % (args.batch, C_FEAT, H_FEAT, H_FEAT)):
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = "stored rocprofv3 --pmc figure (profiles/traffic_corr_argmax.json, FETCH_SIZE x2 + WRITE_SIZE), not re-measured in this run"
    if os.environ.get("IPSR_BENCH_STEP_ONLY", "0") == "1":
        # profiling aid (rocprofv3 of the training steps alone): no layer micro-benchmarks, no CPU twin
        print(json.dumps({"metric": METRIC, "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "note": "IPSR_BENCH_STEP_ONLY=1 (profiling run)"}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    fwd_ms, bwd_ms = layer_timing(device)
    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "ms_per_step_median": round(statistics.median(step_ms), 3),
        "host_enqueue_ms_per_step": round(host_enqueue_ms, 3), "step_graph": graph_info,
        "images_per_sec_median_step": round(args.batch * world / (statistics.median(step_ms) * 1e-3), 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("BASELINE config 2: 256x256 synthetic images, 128x128 centre mask (M=256 of N=1024 "
                                "feature positions), batch %d/GPU, fp32, full IPSR training step" % args.batch)
                   if args.dtype == "f32" else
                   ("BASELINE config 5 (NOT the headline metric): as config 2 with the convolutions under bf16 autocast, "
                    "IPSR layer / InnerCos / losses fp32, batch %d/GPU" % args.batch),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world, "dropout": True,
                   "conv_math": (args.conv_math if args.dtype == "f32" else "bf16x3 (Winograd operands split hi + lo on the bf16 MFMA, fp32 accumulate)"),
                   "vgg16": "seeded random init (no pretrained weights offline)"},
        "ipsr_layer_ms": {"forward": round(fwd_ms, 4), "backward": round(bwd_ms, 4), "total": round(fwd_ms + bwd_ms, 4),
                          "shape": "[%d,%d,%d,%d], M=256" % (BATCH, C_FEAT, H_FEAT, H_FEAT),
                          "inputs": "standalone, synthetic x=|N(0,1)|, ref=relu(N(0,1)), median of 50",
                          "in_step": {"forward": round(statistics.median(fwd_in_step), 4) if fwd_in_step else None,
                                      "backward": round(statistics.median(bwd_in_step), 4) if bwd_in_step else None,
                                      "total": round(statistics.median(fwd_in_step) + statistics.median(bwd_in_step), 4)
                                      if fwd_in_step and bwd_in_step else None,
                                      "calls_timed": [len(fwd_in_step), len(bwd_in_step)],
                                      "inputs": ("HIP events around ipsr_forward / ipsr_backward inside the timed training steps "
                                                 "(real signed conv features, batch %d)" % args.batch) if sgraph is None else
                                                ("HIP events around ipsr_forward / ipsr_backward in %d eager training steps right after the timed "
                                                 "loop, which replayed a HIP graph (real signed conv features, batch %d)" % (gsteps, args.batch))}},
        "ipsr_layer_ms_bf16corr": layer_timing_bf16corr(device),
        "ipsr_layer_ms_other_configs": layer_timing_other_configs(device),
        "roofline": {"kernel": "ipsr::corr_argmax_kernel (fp32 MFMA correlation + arg-max)" if args.dtype == "f32" else
                               "ipsr::corr_argmax_bf16_kernel (correlation + arg-max on v_mfma_f32_32x32x16_bf16, operands rounded to bf16)", "bound": "mfma",
                     "achieved": round(achieved, 3) if achieved else None, "peak": corr_peak,
                     "unit": "TFLOP/s", "frac": round(achieved / corr_peak, 4) if achieved else None,
                     "traffic": traffic, "traffic_source": traffic_src, "flops_per_launch": flops, "kernel_ms": round(kms, 5), "launches_timed": len(kern_ms)},
        # the whole step against the same roofline: direct-convolution FLOPs of the step AS EXECUTED here
        # the convolutions' own matrix-core kernel: every launch of ipsr::wino_gemm_kernel (the 36 GEMMs of the Winograd
        # F(4x4,3x3) / F(3x3,4x4) convolutions: VGG16, netG's 3x3 and dilated 4x4 layers, forward / input / weight gradients)
        # inside the timed steps; flops = the multiplies the kernel executes (4x fewer than the direct convolutions it replaces)
        # `achieved` / `frac`: USEFUL flops (unpadded channels / tiles) over the kernel's time; `executed_*`: the flops the kernel
        # really issues (rows / columns rounded up to the 128 x 128 tile — the difference is arithmetic on zero padding)
        "conv_roofline": conv_roofline(args, ng, gsteps, gemm_ms, gemm_flops, gemm_useful),
        "conv_roofline_direct_bf16": direct,
        "step_roofline": {"bound": "mfma", "flops_per_image": STEP_FLOPS_PER_IMAGE,
                          "achieved_direct_equivalent": round(STEP_FLOPS_PER_IMAGE * value / world / 1e12, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
                          "unit": "TFLOP/s per GPU",
                          "direct_equivalent_frac": round(STEP_FLOPS_PER_IMAGE * value / world / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                          "note": "NOT a hardware fraction: the DIRECT-convolution flop count of the step as executed, over time.  The Winograd "
                                  "kernels (this repo's F(4x4,3x3) / F(3x3,4x4) / F(5x5,2x2), MIOpen's F(2x2,3x3)) execute 2.25-4x fewer "
                                  "multiplies than this count, so the figure could exceed 1"}
        if args.dtype == "f32" else None,
        "alt_arithmetic": alt,
        "new_mask_each_step": newmask,
        "innercos_roofline": ic_step,
        "strict_reference": strict,
        "ddp": ddp,
        "losses_last_step": {k: round(v, 4) for k, v in errs.items()},
        "losses_last_timed_step": {k: round(v, 4) for k, v in timed_losses.items()}, "losses_finite": losses_finite,
    }
    if world == 1 and not args.no_cpu_baseline:
        print("[bench] GPU part done: %.2f images/s; timing the CPU twin ..." % value, file=sys.stderr, flush=True)
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
