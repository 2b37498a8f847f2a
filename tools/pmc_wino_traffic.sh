# HBM traffic of the Winograd convolution's kernels on 512 -> 512 @32x32, batch 8 (FETCH_SIZE / WRITE_SIZE of the wino_* kernels)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass --kernel-trace -d gpurun_out/pmc_$pass -o p --output-format csv -- python tools/exp_transforms.py --iters 5 --math fp32 > gpurun_out/pmc_$pass.log 2>&1
done
python - <<'PY'
import csv, glob, collections
out = {}
for p in ("FETCH_SIZE", "WRITE_SIZE"):
    f = (glob.glob("gpurun_out/pmc_%s/*counter_collection.csv" % p) + glob.glob("gpurun_out/pmc_%s/*/*counter_collection.csv" % p))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != p: continue
        key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r.get("Workgroup_Size", ""))
        agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
    out[p] = agg
keys = sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]))
with open("gpurun_out/r3_wino_hbm_traffic.csv", "w") as fh:
    fh.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python tools/exp_transforms.py --iters 5 --math fp32; per launch; FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads); KB units as reported x 1024\n")
    fh.write("kernel,grid,launches,fetch_MB_per_launch(x2),write_MB_per_launch,total_MB\n")
    for k in keys:
        f = out["FETCH_SIZE"].get(k, [0, 0.0]); w = out["WRITE_SIZE"].get(k, [0, 0.0])
        fm = 2 * f[1] / max(f[0], 1) * 1024 / 1e6; wm = w[1] / max(w[0], 1) * 1024 / 1e6
        fh.write("%s,%s,%d,%.2f,%.2f,%.2f\n" % (k[0], k[1], max(f[0], w[0]), fm, wm, fm + wm))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
