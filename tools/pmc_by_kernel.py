#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs) -> HBM bytes per launch, per (kernel, grid), corrected as
MI355X_MICROARCH.md prescribes (both counters in KiB; FETCH_SIZE x2 on gfx950 for wide coalesced reads).

    python tools/pmc_by_kernel.py gpurun_out/pmc_f/x_counter_collection.csv gpurun_out/pmc_w/x_counter_collection.csv [substring ...]
"""
import collections
import csv
import sys


def per_kernel(path, counter, subs):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not any(s in name for s in subs):
            continue
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        agg[(name.split("(")[0].replace("void ", ""), grid)].append(float(r["Counter_Value"]))
    return agg


def main():
    subs = sys.argv[3:] or ["ipsr::"]
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE", subs), per_kernel(sys.argv[2], "WRITE_SIZE", subs)
    print("kernel,grid,launches,fetch_MB_per_launch(x2),write_MB_per_launch,total_MB")
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, [0.0]), w.get(k, [0.0])
        fm, wm = 2.0 * sum(fv) / len(fv) * 1024 / 1e6, sum(wv) / len(wv) * 1024 / 1e6
        print("%s,%s,%d,%.2f,%.2f,%.2f" % (k[0], k[1], max(len(fv), len(wv)), fm, wm, fm + wm))


if __name__ == "__main__":
    main()
