#!/usr/bin/env python3
"""Which module of a net is the first whose forward output differs between two passes on the same input and weights?

    B=8 python tools/exp_first_divergence.py [netP|netG|netD|netF] [--bf16]
"""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402

B = int(os.environ.get("B", "8"))
which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "netP"
bf16 = "--bf16" in sys.argv
torch.backends.cudnn.deterministic = True
torch.manual_seed(100)
opt = Option(gpu_ids=[0], batchSize=B, use_dropout=False, quiet=True, allow_random_vgg=True, amp_bf16=bf16, checkpoints_dir="/tmp/ck_div")
with contextlib.redirect_stdout(io.StringIO()):
    m = create_model(opt)
g = torch.Generator(device="cuda").manual_seed(7)
img = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
ref = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
mask[:, :, 64:192, 64:192] = 1
net = getattr(m, which)
names = {mod: name for name, mod in net.named_modules()}
record = []


def hook(mod, inp, out):
    if torch.is_tensor(out):
        eng = getattr(mod, "last_engine", None) or getattr(mod, "_last_engine", None)
        record[-1].append((names.get(mod, "?"), type(mod).__name__, tuple(out.shape), eng, out.detach().clone()))


for mod in net.modules():
    mod.register_forward_hook(hook)
for r in range(3):
    record.append([])
    m.set_input(img, mask, ref)
    m.set_ref_latent()
    m.set_gt_latent()
    with torch.no_grad():
        m.forward()
        if which in ("netD", "netF"):
            m.backward_D.__func__  # (forward of the discriminators happens inside backward_D)
    if which in ("netD", "netF"):
        m.backward_D()
torch.cuda.synchronize()
for r in (1, 2):
    first = None
    nd = 0
    for a, b in zip(record[0], record[r]):
        if not torch.equal(a[4], b[4]):
            nd += 1
            if first is None:
                first = (a[0], a[1], a[2], a[3], float((a[4].float() - b[4].float()).abs().max()))
    print("run 0 vs run %d: %d of %d module outputs differ; first: %s" % (r, nd, len(record[0]), first), flush=True)
