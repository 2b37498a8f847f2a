#!/usr/bin/env python3
"""Per-kernel means of an MFMA PMC pass (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv): mfma_util = MFMA_BUSY / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)  (the counters
come summed over the 8 XCDs; MI355X_MICROARCH.md "DVFS give-back").

    python tools/collect_mfma.py gpurun_out/pmc_conv/*counter_collection.csv > profiles/r02_conv_mfma_pmc.csv
"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "ipsr::" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "GRBM_GUI_ACTIVE", "mfma_util"])
for k in sorted(agg):
    c = agg[k]
    m = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else 0.0
    busy, act = m("SQ_VALU_MFMA_BUSY_CYCLES"), m("GRBM_GUI_ACTIVE")
    w.writerow([k, len(next(iter(c.values()))), int(busy), int(m("SQ_INSTS_VALU_MFMA_MOPS_F32")), int(act),
                round(busy / (1024.0 * act / 8.0), 4) if act else 0.0])
