set -e
python -m pytest tests/test_gpu_model.py -q -k "fused or instnorm" > gpurun_out/r3_in.log 2>&1 || { tail -20 gpurun_out/r3_in.log; exit 1; }
tail -2 gpurun_out/r3_in.log
export IPSR_BENCH_NO_ALT=1
for rep in 1 2; do
python bench.py --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('planes to 65536', d['value'],d['ms_per_step_median'])"
python -c "
import deepinpainting_amd.ops as o; o.INSTNORM_MAX_PLANE=16384
import runpy,sys; sys.argv=['bench.py','--steps','30','--warmup','5']; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('planes to 16384', d['value'],d['ms_per_step_median'])"
done
