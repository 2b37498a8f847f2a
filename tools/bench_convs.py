#!/usr/bin/env python3
"""Per-layer conv table of the training step: every distinct Conv2d / ConvTranspose2d the four nets
execute at BASELINE config 2 (batch 8, 256x256, fp32), timed forward / backward-data / backward-weight on MIOpen with the
shipped find-db, with direct-convolution FLOPs -> TFLOP/s.  Shows where the 157 TF fp32 MFMA roofline is being missed.

    python tools/bench_convs.py [--batch 8] > gpurun_out/convs.txt
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa  (private MIOpen db copy)
from deepinpainting_amd.options import Option
from deepinpainting_amd.models.models import create_model

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()
B = args.batch

opt = Option(gpu_ids=[0], batchSize=B, quiet=True, allow_random_vgg=True, checkpoints_dir="/tmp/ipsr_ck")
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    m = create_model(opt)

records = []      # (net, name, module, in_shape)
def hook(net, name):
    def f(mod, inp, out):
        records.append((net, name, mod, tuple(inp[0].shape), tuple(out.shape)))
    return f
hs = []
for net in ("netP", "netG", "netD", "netF", "vgg"):
    for name, mod in getattr(m, net).named_modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            hs.append(mod.register_forward_hook(hook(net, name)))
g = torch.Generator(device="cuda").manual_seed(1)
img = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
ref = torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1
mask = torch.zeros(1, 1, 256, 256, dtype=torch.bool, device="cuda"); mask[..., 64:192, 64:192] = True
m.set_input(img, mask, ref); m.set_ref_latent(); m.set_gt_latent(); m.optimize_parameters()
for h in hs: h.remove()
torch.cuda.synchronize()

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

# group identical configurations; count executions per step (forward calls; trainable nets also run backward)
groups = {}
for net, name, mod, ish, osh in records:
    tr = isinstance(mod, torch.nn.ConvTranspose2d)
    key = (tr, ish, mod.weight.shape, mod.stride, mod.padding, mod.dilation)
    gr = groups.setdefault(key, {"mod": mod, "osh": osh, "calls": 0, "where": set(), "train": net != "vgg"})
    gr["calls"] += 1; gr["where"].add(net)

tot = {"fwd": 0.0, "bwd": 0.0}
print("%-5s %-22s %-18s %-9s %5s | %8s %6s | %8s %6s | %8s %6s | %s" % ("kind", "input", "weight", "s/p/d", "calls", "fwd ms", "TF", "bwdD ms", "TF", "bwdW ms", "TF", "nets"))
rows = []
for key, gr in groups.items():
    tr, ish, wsh, stride, pad, dil = key
    mod = gr["mod"]; osh = gr["osh"]
    x = torch.randn(ish, device="cuda"); w = mod.weight.detach()
    go = torch.randn(osh, device="cuda")
    if tr:
        fl = 2.0 * ish[0] * ish[1] * ish[2] * ish[3] * wsh[1] * wsh[2] * wsh[3]
    else:
        fl = 2.0 * osh[0] * osh[1] * osh[2] * osh[3] * wsh[1] * wsh[2] * wsh[3]
    with torch.no_grad():
        if tr:
            f = lambda: torch.nn.functional.conv_transpose2d(x, w, None, stride, pad, 0, 1, dil)
        else:
            f = lambda: torch.nn.functional.conv2d(x, w, None, stride, pad, dil)
        tf = t(f)
        cb = torch.ops.aten.convolution_backward
        args_ = (go, x, w, None, list(stride), list(pad), list(dil), tr, [0, 0], 1)
        td = t(lambda: cb(*args_, [True, False, False]))
        tw = t(lambda: cb(*args_, [False, True, False]))
    rows.append((fl * gr["calls"], tr, ish, wsh, stride, pad, dil, gr, tf, td, tw, fl))
rows.sort(key=lambda r: -(r[8] + r[9] + r[10]) * r[7]["calls"])
sf = sd = sw = 0.0
for _, tr, ish, wsh, stride, pad, dil, gr, tf, td, tw, fl in rows:
    c = gr["calls"]
    print("%-5s %-22s %-18s %-9s %5d | %8.3f %6.1f | %8.3f %6.1f | %8.3f %6.1f | %s" % (
        "convT" if tr else "conv", "x".join(map(str, ish)), "x".join(map(str, wsh)), "%d/%d/%d" % (stride[0], pad[0], dil[0]), c,
        tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9, ",".join(sorted(gr["where"]))))
    sf += tf * c
    if gr["train"]:
        sd += td * c; sw += tw * c
print("sum over calls: fwd %.2f ms, bwd-data %.2f ms (upper bound: trainable nets, every call), bwd-weight %.2f ms" % (sf, sd, sw))
