"""How repeatable are two backward passes of the trainer from identical weights and inputs (one process, no exchange)?
Prints, per net, the relative L2 distance of the whole gradient between run 0 and runs 1, 2 — with and without
torch.backends.cudnn.deterministic (MIOpen: no atomically accumulated weight gradients)."""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402,F401
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402

B = int(os.environ.get("B", "1"))
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    torch.manual_seed(100)
    opt = Option(gpu_ids=[0], batchSize=B, use_dropout=False, quiet=True, allow_random_vgg=True, checkpoints_dir="/tmp/ck_rep")
    with contextlib.redirect_stdout(io.StringIO()):
        m = create_model(opt)
    g = torch.Generator(device="cuda").manual_seed(7)
    img = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    ref = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda")
    mask[:, :, 64:192, 64:192] = 1
    nets = (("G", m.netG), ("P", m.netP), ("D", m.netD), ("F", m.netF))
    runs = []
    for r in range(3):
        m.set_input(img, mask, ref)
        m.set_ref_latent()
        m.set_gt_latent()
        m.forward()
        for _, net in nets:
            for p in net.parameters():
                p.grad = None
        m.backward_D()
        m.backward_G()
        runs.append(({t: torch.cat([p.grad.reshape(-1) for p in n.parameters() if p.grad is not None]).clone() for t, n in nets},
                     m.fake_B.detach().clone(), m.fake_P.detach().clone()))
    for r in (1, 2):
        print("deterministic=%s run0 vs run%d: fake_B equal %s, fake_P equal %s; " % (det, r, torch.equal(runs[0][1], runs[r][1]), torch.equal(runs[0][2], runs[r][2])) +
              ", ".join("%s %.2e" % (t, float((runs[0][0][t] - runs[r][0][t]).double().norm() / runs[0][0][t].double().norm())) for t, _ in nets), flush=True)
