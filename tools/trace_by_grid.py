import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[int(r["Start_Timestamp"]) for r in rows if "corr_argmax_fast_kernel" in r["Kernel_Name"]]
steps=[m for i,m in enumerate(marks) if i+1<len(marks) and marks[i+1]-m>5e6]
t0=steps[3]; t1=steps[-1]; nst=len(steps)-1-3
agg=collections.defaultdict(list)
for r in rows:
    ts=int(r["Start_Timestamp"])
    if ts<t0 or ts>=t1: continue
    n=r["Kernel_Name"]
    if "ipsr::wino" not in n or "gemm" in n: continue
    key=(n.split("(")[0].replace("void ",""), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key].append(int(r["End_Timestamp"])-ts)
out=[]
for k,v in agg.items():
    out.append((sum(v)/nst/1e3, k, len(v)/nst, sum(v)/len(v)/1e3))
out.sort(reverse=True)
for tot,k,n,avg in out[:60]:
    print("%8.1f us/step  %-45s grid %6s x %5s x %s  n/step %4.1f  avg %7.1f us"%(tot,k[0][:45],k[1],k[2],k[3],n,avg))
