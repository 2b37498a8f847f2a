#!/usr/bin/env python3
"""Experiment: do two independent convolution pipelines overlap when issued on two HIP streams?  (input gradient and weight
gradient of one layer are independent; the transforms of one can run under the GEMM of the other.)

    python tools/exp_streams.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinpainting_amd import ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = 8
    side = torch.cuda.Stream()
    for (Cin, H, Cout) in ((512, 32, 512), (256, 64, 256), (128, 128, 128), (1024, 32, 256)):
        x = torch.randn(B, Cin, H, H, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        dy = torch.randn(B, Cout, H, H, device="cuda")

        def dx():
            return ops.conv3x3_winograd(ops.CONV_BWD_DATA, dy, w, (B, Cin, H, H), Cout)

        def dw():
            return ops.conv3x3_winograd_wrw(False, x, dy, Cout)

        def serial():
            dx(); dw()

        def two():
            main_s = torch.cuda.current_stream()
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                dw()
            dx()
            main_s.wait_stream(side)
        t_dx, t_dw, t_ser, t_two = timeit(dx), timeit(dw), timeit(serial), timeit(two)
        print("%4d->%4d @%3d  dx %.3f  dw %.3f  serial %.3f  two streams %.3f ms  (%.0f %% of serial)" %
              (Cin, Cout, H, t_dx, t_dw, t_ser, t_two, 100 * t_two / t_ser), flush=True)


if __name__ == "__main__":
    main()
