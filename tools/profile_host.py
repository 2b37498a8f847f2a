#!/usr/bin/env python3
"""Where does the host spend a training step's queueing time?  cProfile over a few un-synchronised steps (bench.py's trainer and
batch), top functions by own time and by cumulative time.

    python tools/profile_host.py [--dtype bf16 --batch 16] > profiles/r04_host_profile_bf16.txt
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from deepinpainting_amd.models.models import create_model
    from deepinpainting_amd.options import Option
    dev = torch.device("cuda", 0)
    opt = Option(gpu_ids=[0], batchSize=a.batch, use_dropout=True, quiet=True, allow_random_vgg=True, amp_bf16=a.dtype == "bf16",
                 checkpoints_dir="/tmp/ipsr_host_profile")
    torch.manual_seed(1234)
    model = bench.quiet(create_model, opt)
    img, mask, ref = bench.synthetic_batch(dev, a.batch, 1234)
    for _ in range(5):
        bench.train_step(model, img, mask, ref)
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bench.train_step(model, img, mask, ref)
    t_plain = (time.perf_counter() - t0) / a.steps * 1e3
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        bench.train_step(model, img, mask, ref)
    pr.disable()
    torch.cuda.synchronize()
    print("host time per step without the profiler: %.2f ms (%s, batch %d)" % (t_plain, a.dtype, a.batch))
    for key, n in (("tottime", 45), ("cumtime", 60)):
        s = io.StringIO()
        st = pstats.Stats(pr, stream=s)
        st.strip_dirs().sort_stats(key).print_stats(n)
        txt = s.getvalue()
        print("==== by %s (all %d steps) ====" % (key, a.steps))
        print(txt[txt.index("ncalls"):] if "ncalls" in txt else txt)


if __name__ == "__main__":
    main()
