#!/usr/bin/env python3
"""Per-step wall times of the training step over a long run (diagnostic: does the step time drift?)."""
import contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (sets the private MIOpen db path)
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402

opt = Option(gpu_ids=[0], batchSize=8, use_dropout=True, quiet=True, allow_random_vgg=True, checkpoints_dir="/tmp/ipsr_trend_ck")
torch.manual_seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    model = create_model(opt)
img, mask, ref = bench.synthetic_batch(torch.device("cuda", 0), 8, 1234)
for _ in range(5):
    bench.train_step(model, img, mask, ref)
torch.cuda.synchronize()
import gc
_gc_t = {}
def _cb(phase, info):
    if phase == 'start':
        _gc_t['t'] = time.perf_counter()
    elif info['generation'] >= 2:
        print('gc gen%d collected=%d took %.1f ms' % (info['generation'], info['collected'], (time.perf_counter() - _gc_t['t']) * 1e3))
gc.callbacks.append(_cb)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    t0 = time.perf_counter()
    bench.train_step(model, img, mask, ref)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
errs = model.get_current_errors()
print("median %.2f ms  max %.2f ms  losses %s" % (sorted(ts)[len(ts) // 2], max(ts), {k: round(v, 3) for k, v in errs.items()}))
print("mem allocated %.2f GB reserved %.2f GB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
