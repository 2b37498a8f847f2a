# Kernel timeline of the last forward+backward of tools/profile_layer.py under rocprofv3 (GPU box): bash tools/layer_timeline.sh [args of profile_layer.py]
export TMPDIR=/tmp
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace -d gpurun_out/prof_tl -o layer --output-format csv -- python3 tools/profile_layer.py "$@" > gpurun_out/layer_timeline_run.txt 2>&1
python3 - <<PY
import csv,glob
f=(glob.glob("gpurun_out/prof_tl/*kernel_trace.csv")+glob.glob("gpurun_out/prof_tl/*/*kernel_trace.csv"))[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
last=max(i for i,n in enumerate(names) if "patch_normalize" in n or "window_prepare" in n)
t0=int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%8.1f +%7.1f us  %s  grid %s"%((s-t0)/1e3,(e-s)/1e3,r["Kernel_Name"].split("(")[0][-60:],r.get("Grid_Size_X","")))
PY
rm -rf gpurun_out/prof_tl
