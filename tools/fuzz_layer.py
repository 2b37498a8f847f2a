#!/usr/bin/env python3
"""Randomised parity sweep of the IPSR layer (HIP through the C-ABI vs the CPU oracle, bit-exact), beyond the fixed cases of
tests/test_gpu_parity.py: random batch / channels / feature size / shift_sz / mask density / feature sign.

    python tools/fuzz_layer.py [--n 60] [--seed 0]
"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops
from oracle import ipsr_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rs = np.random.RandomState(a.seed)
bad = 0
t0 = time.time()
for it in range(a.n):
    p = int(rs.choice([1, 1, 1, 2, 3]))
    B = int(rs.randint(1, 4))
    C = int(rs.choice([3, 8, 12, 16, 20, 32, 64, 96, 128, 200, 256, 512]))
    h, w = int(rs.randint(p, 22)), int(rs.randint(p, 22))
    if p > 1 and C > 128:                   # wide patches (C*p*p up to 4608: the multi-wave recurrence): keep the oracle fast
        h, w = int(rs.randint(p, 11)), int(rs.randint(p, 11))
    if p == 1 and rs.rand() < 0.2:
        h = w = int(rs.choice([16, 32]))
    signed = rs.rand() < 0.3
    dens = float(rs.choice([0.0, 0.05, 0.2, 0.5, 1.0]))
    x = rs.standard_normal((B, C, h, w)).astype(np.float32)
    if not signed:
        x = np.abs(x)
    ref = (rs.standard_normal((B, C, h, w)) if signed else rs.rand(B, C, h, w)).astype(np.float32)
    feat = (rs.rand(h, w) < dens).astype(np.uint8)
    mpi = orc.index_prep(feat, patch=p).mask_point_idx
    g = rs.standard_normal((B, C, h, w)).astype(np.float32)
    tw = float(rs.choice([0.0, 0.5, 1.0]))
    fo = orc.forward(x, ref, mpi, patch=p)
    go = orc.backward_patch(g, len(mpi), fo.bwd_index, tw, p)
    d = lambda t, dt=None: torch.from_numpy(np.ascontiguousarray(t)).cuda() if dt is None else torch.from_numpy(np.ascontiguousarray(t)).to(dt).cuda()
    f = ops.forward(d(x), d(ref), d(mpi, torch.int32), patch=p, want_attn=True)
    gin = ops.backward(d(g), f.bwd_index, tw, len(mpi), patch=p)
    torch.cuda.synchronize()
    ok = (np.array_equal(f.ind.cpu().numpy(), fo.ind) and np.array_equal(f.vmax.cpu().numpy(), fo.vmax)
          and np.array_equal(f.attn_rows.cpu().numpy(), fo.attn_rows) and np.array_equal(f.out.cpu().numpy(), fo.out)
          and np.array_equal(gin.cpu().numpy(), go))
    finite = np.isfinite(fo.out).all()
    if not ok:
        # non-finite oracle values (signed features can drive at + vmax to 0) compare unequal as NaN != NaN: check by bits
        ok = (np.array_equal(f.out.cpu().numpy().view(np.int32) if not finite else f.out.cpu().numpy(), fo.out.view(np.int32) if not finite else fo.out))
    if not ok:
        bad += 1
        print("MISMATCH it=%d p=%d B=%d C=%d %dx%d M=%d signed=%s dens=%.2f tw=%.1f finite=%s" % (it, p, B, C, h, w, len(mpi), signed, dens, tw, finite), flush=True)
print("fuzz: %d cases, %d mismatches, %.1f s" % (a.n, bad, time.time() - t0))
sys.exit(1 if bad else 0)
