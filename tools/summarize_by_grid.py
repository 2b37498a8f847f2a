#!/usr/bin/env python3
"""Per (kernel, grid) totals of the steady-state training steps of a rocprofv3 --kernel-trace CSV (see summarize_trace.py for
how the steady state is cut out).

    python tools/summarize_by_grid.py gpurun_out/prof/x_kernel_trace.csv --match wino > profiles/xyz_by_grid.txt
"""
import argparse
import collections
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="corr_argmax_fast_kernel")
    ap.add_argument("--skip", type=int, default=3)
    ap.add_argument("--match", default="", help="only kernels whose name contains this")
    ap.add_argument("--top", type=int, default=80)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if a.marker in r["Kernel_Name"]]
    steps = [m for i, m in enumerate(marks) if i + 1 < len(marks) and marks[i + 1] - m > 5e6]
    if len(steps) <= a.skip + 1:
        sys.exit("not enough marker launches (%d)" % len(steps))
    t_lo, t_hi = steps[a.skip], steps[-1]
    n = len(steps) - 1 - a.skip
    agg = collections.OrderedDict()
    for r in rows:
        if not (t_lo <= int(r["Start_Timestamp"]) < t_hi) or a.match not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0]
        key = (name, r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"))
        e = agg.setdefault(key, [0, 0])
        e[0] += 1
        e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for (name, gx, gy, gz, wx), e in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print("%8.1f us/step  %-52s grid %7s x %5s x %s (wg %s)  n/step %5.1f  avg %7.1f us" %
              (e[1] / n / 1e3, name[-52:], gx, gy, gz, wx, e[0] / n, e[1] / e[0] / 1e3))


if __name__ == "__main__":
    main()
