#!/usr/bin/env python3
"""How many attention entries survive the reference's LongTensor truncation during real training steps?"""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepinpainting_amd import ops  # noqa: E402
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402

opt = Option(gpu_ids=[0], batchSize=8, use_dropout=True, quiet=True, allow_random_vgg=True, checkpoints_dir="/tmp/ipsr_surv_ck")
torch.manual_seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    model = create_model(opt)
img, mask, ref = bench.synthetic_batch(torch.device("cuda", 0), 8, 1234)
orig = ops.forward
last = {}
def spy(*a, **k):
    f = orig(*a, **k)
    last["f"] = f
    last["x"] = a[0]
    return f
ops.forward = spy
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    bench.train_step(model, img, mask, ref)
    f = last["f"]
    N, M = 1024, 256
    bi = f.bwd_index.cpu()
    offB = bi[:, 2 * N + 1:3 * N + 2]
    tot = offB[:, N]
    lens = (offB[:, 1:] - offB[:, :-1])
    x = last["x"]
    print("step %2d: survivors/sample %s  max column length %d  |x| max %.2f  frac(x<0) %.2f" %
          (step, tot.tolist(), int(lens.max()), float(x.abs().max()), float((x < 0).float().mean())))
