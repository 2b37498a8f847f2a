#!/usr/bin/env python3
"""Loss trajectory of the trainer on one fixed synthetic batch, hand-written convolution engines vs MIOpen only
(IPSR_CONV_ENGINE): the engines change fp32 rounding (Winograd ~2e-5 per layer), not the optimisation.

    python tools/loss_trend.py [steps]            # prints G_GAN / G_L1 / D / F every 10 steps for both settings
"""
import contextlib, io, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepinpainting_amd.models import hipconv  # noqa: E402
from deepinpainting_amd.models.models import create_model  # noqa: E402
from deepinpainting_amd.options import Option  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
# (label, conv engine, conv_math for fp32 activations, bf16 autocast)
RUNS = [("auto", "auto", "fp32", False), ("miopen", "miopen", "fp32", False)]
if len(sys.argv) > 2 and sys.argv[2] == "math":          # the split-bf16 arithmetics and BASELINE config 5 against the fp32 run
    RUNS = [("fp32", "auto", "fp32", False), ("bf16x6", "auto", "bf16x6", False), ("bf16x3", "auto", "bf16x3", False), ("amp-bf16", "auto", "fp32", True)]
for engine, force, math, amp in RUNS:
    hipconv._FORCE = force
    opt = Option(gpu_ids=[0], batchSize=8, use_dropout=False, quiet=True, allow_random_vgg=True, checkpoints_dir="/tmp/ipsr_loss_ck",
                 conv_math=math, amp_bf16=amp)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        model = create_model(opt)
    img, mask, ref = bench.synthetic_batch(torch.device("cuda", 0), 8, 1234)
    for step in range(steps + 1):
        bench.train_step(model, img, mask, ref)
        if step % 10 == 0:
            e = model.get_current_errors()
            vals = {k: float(v) for k, v in e.items()}
            bad = any(v != v or abs(v) == float("inf") for v in vals.values())
            print("%-8s step %3d  " % (engine, step) + "  ".join("%s %.4f" % (k, v) for k, v in vals.items()) + ("  NON-FINITE" if bad else ""), flush=True)
    del model
    torch.cuda.empty_cache()
hipconv._FORCE = None
