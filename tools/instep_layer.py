#!/usr/bin/env python3
"""Capture the IPSR layer's REAL inputs inside training steps (signed conv features) and report, per sample, the structure
of the sparse trunc(kbar) the backward walks (one-hot part A, truncation survivors B), then time ipsr_forward /
ipsr_backward stand-alone on exactly those tensors (median of 30, HIP events).

    python tools/instep_layer.py [steps] > gpurun_out/instep_layer.txt
"""
import contextlib, io, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepinpainting_amd import ops  # noqa: E402
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402

opt = Option(gpu_ids=[0], batchSize=8, use_dropout=True, quiet=True, allow_random_vgg=True, checkpoints_dir="/tmp/ipsr_instep_ck")
torch.manual_seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    model = create_model(opt)
img, mask, ref = bench.synthetic_batch(torch.device("cuda", 0), 8, 1234)
orig_f, orig_b = ops.forward, ops.backward
cap = {}
def spy_f(*a, **k):
    f = orig_f(*a, **k)
    cap["x"], cap["ref"], cap["mpi"], cap["f"] = a[0].clone(), a[1].clone(), a[2].clone(), f
    cap["args"], cap["kw"] = a[3:], k
    return f
def spy_b(g, bidx, tw, M, patch=1):
    cap["g"], cap["tw"], cap["M"] = g.clone(), tw, M
    return orig_b(g, bidx, tw, M, patch)
ops.forward, ops.backward = spy_f, spy_b
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for step in range(steps):
    bench.train_step(model, img, mask, ref)
ops.forward, ops.backward = orig_f, orig_b
torch.cuda.synchronize()
x, rf, mpi, g, M = cap["x"], cap["ref"], cap["mpi"], cap["g"], cap["M"]
B, C, h, w = x.shape
N = h * w
bi = cap["f"].bwd_index.cpu()
offA = bi[:, :N + 1]
offB = bi[:, 2 * N + 1:3 * N + 2]
la = offA[:, 1:] - offA[:, :-1]
lb = offB[:, 1:] - offB[:, :-1]
print("after %d training steps: x [%d,%d,%d,%d], M=%d, |x| max %.2f, frac(x<0) %.2f" % (steps, B, C, h, w, M, float(x.abs().max()), float((x < 0).float().mean())))
print("%-6s | %-44s | %-52s" % ("sample", "A: one-hot lists (N-M entries)", "B: truncation survivors"))
for b in range(B):
    a, bb = la[b], lb[b]
    print("%-6d | max %4d  cols>8: %3d (entries %5d)  cols>64: %3d | total %6d  max %4d  cols>8: %3d (entries %6d)  cols>64: %3d" % (
        b, int(a.max()), int((a > 8).sum()), int(a[a > 8].sum()), int((a > 64).sum()),
        int(bb.sum()), int(bb.max()), int((bb > 8).sum()), int(bb[bb > 8].sum()), int((bb > 64).sum())))
tot = la + lb
hist = torch.bincount(torch.clamp(tot.reshape(-1), max=300) // 10, minlength=31)
print("combined column length histogram (bins of 10, all samples):", hist.tolist())

def timed(fn, n=30):
    """Kernel time through the C-ABI's own HIP-event hook (regions 1 = ipsr_forward, 2 = ipsr_backward): wall time around a
    single call would mostly measure the host (allocation of the outputs, launch latency) at these durations."""
    from deepinpainting_amd import _lib
    import ctypes
    lib = _lib.lib()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.ipsr_profile_enable(n)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    out = []
    buf = (ctypes.c_float * 4096)()
    for region in (1, 2):
        k = lib.ipsr_profile_read_region(region, ctypes.cast(buf, ctypes.c_void_p), n)
        if k > 0:
            out.append(statistics.median(list(buf[:k])))
    lib.ipsr_profile_enable(0)
    return out[0] if out else float("nan")

fw = lambda xx, rr: ops.forward(xx, rr, mpi, *cap["args"], **cap["kw"])      # same index / counts / options as the step's own call
f = fw(x, rf)
print("stand-alone on these tensors: forward %.4f ms, backward %.4f ms" % (timed(lambda: fw(x, rf)), timed(lambda: ops.backward(g, f.bwd_index, cap["tw"], M))))
gs = torch.randn_like(x).abs()
fs = fw(gs, torch.relu(torch.randn_like(x)))
print("synthetic non-negative features: forward %.4f ms, backward %.4f ms" % (timed(lambda: fw(gs, rf.abs())), timed(lambda: ops.backward(g, fs.bwd_index, 1.0, M))))
if len(sys.argv) > 2:
    torch.save({"x": x.cpu(), "ref": rf.cpu(), "mpi": mpi.cpu(), "g": g.cpu()}, sys.argv[2])

