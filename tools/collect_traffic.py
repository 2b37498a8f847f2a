#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — separate runs, they do not fit one pass on gfx950)
into per-kernel HBM traffic, corrected as MI355X_MICROARCH.md §HBM prescribes:
  * both counters are in KiB;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> x2;
  * WRITE_SIZE is exact for 16-B-per-lane streaming stores (other widths uncalibrated).

    python tools/collect_traffic.py gpurun_out/pmc_fetch/l_counter_collection.csv \
                                    gpurun_out/pmc_write/l_counter_collection.csv > profiles/r01_layer_cfg2_hbm_traffic.csv
Also rewrites profiles/traffic_corr_argmax.json, which bench.py reports as roofline.traffic.
"""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "ipsr::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    w = csv.writer(sys.stdout)
    w.writerow(["Kernel", "FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB_raw", "hbm_read_bytes(2x corrected)", "hbm_write_bytes", "hbm_bytes_per_launch"])
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, wr = fetch.get(k, 0.0), write.get(k, 0.0)
        rd_b, wr_b = 2.0 * f * 1024.0, wr * 1024.0
        out[k] = rd_b + wr_b
        w.writerow([k, round(f, 1), round(wr, 1), int(rd_b), int(wr_b), int(rd_b + wr_b)])
    key = [k for k in out if "corr_argmax" in k]
    if key:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with open(os.path.join(root, "profiles", "traffic_corr_argmax.json"), "w") as fh:
            json.dump({"kernel": key[0], "hbm_bytes_per_launch": int(out[key[0]]),
                       "fetch_size_kib_raw": fetch.get(key[0]), "write_size_kib_raw": write.get(key[0]),
                       "correction": "FETCH_SIZE x2 (gfx950 half-count for 16 B/lane streams), both counters x1024",
                       "workload": "BASELINE config 2 layer shape [8,512,32,32]",
                       "algorithmic_bytes": 2 * 8 * 512 * 1024 * 4 + 2 * 8 * 1024 * 4}, fh, indent=1)


if __name__ == "__main__":
    main()
