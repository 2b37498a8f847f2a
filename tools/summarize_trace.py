#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into per-kernel stats of the STEADY STATE.

MIOpen spends the first training step in its one-off solver search (naive_conv_* kernels, tens of seconds);
those dispatches would drown the numbers that matter.  The steady state is taken as everything from the
`--skip`-th launch of the marker kernel (default: the IPSR correlation kernel, one launch per training step).

    python tools/summarize_trace.py gpurun_out/prof_bench/bench_kernel_trace.csv --skip 3 > profiles/xyz.csv
"""
import argparse
import collections
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="corr_argmax_fast_kernel")
    ap.add_argument("--skip", type=int, default=3, help="marker launches to skip (warm-up steps)")
    ap.add_argument("--steps", type=int, default=0, help="marker launches to include (0 = all remaining training steps)")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if a.marker in r["Kernel_Name"]]
    # the layer microbenchmark at the end of bench.py launches the marker back to back: keep only launches
    # that are at least 5 ms apart (training steps)
    steps = [m for i, m in enumerate(marks) if i + 1 < len(marks) and marks[i + 1] - m > 5e6]
    if len(steps) <= a.skip + 1:
        sys.exit("not enough marker launches (%d)" % len(steps))
    t_lo = steps[a.skip]
    t_hi = steps[a.skip + a.steps] if a.steps and a.skip + a.steps < len(steps) else steps[-1]
    nsteps = steps.index(t_hi) - a.skip
    sel = [r for r in rows if t_lo <= int(r["Start_Timestamp"]) < t_hi]
    agg = collections.OrderedDict()
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
    tot = sum(e[1] for e in agg.values())
    wall = t_hi - t_lo
    w = csv.writer(sys.stdout)
    w.writerow(["# steady state: %d training steps, wall %.3f ms/step, kernel time %.3f ms/step (GPU busy %.1f%%), %d dispatches/step"
                % (nsteps, wall / nsteps / 1e6, tot / nsteps / 1e6, 100.0 * tot / wall, len(sel) // nsteps)])
    w.writerow(["Name", "CallsPerStep", "AverageNs", "MinNs", "MaxNs", "TotalPerStepUs", "Percentage"])
    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    keep = ranked[:a.top] + [kv for kv in ranked[a.top:] if "ipsr::" in kv[0]]      # this repo's own kernels are always listed
    for name, e in keep:
        w.writerow([name[:160], round(e[0] / nsteps, 2), round(e[1] / e[0], 1), e[2], e[3], round(e[1] / nsteps / 1e3, 2), round(100.0 * e[1] / tot, 2)])


if __name__ == "__main__":
    main()
