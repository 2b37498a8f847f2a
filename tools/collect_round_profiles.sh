# Round profiles: steady-state kernel summary of the training step, per-(kernel, grid) totals, layer kernel summaries, final bench lines.
# usage (GPU box): bash tools/collect_round_profiles.sh r03
R=${1:-rXX}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
IPSR_BENCH_STEP_ONLY=1 rocprofv3 --kernel-trace -d gpurun_out/prof_step -o step --output-format csv -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/${R}_prof_step.json 2> gpurun_out/${R}_prof_step.err
F=$(ls gpurun_out/prof_step/*kernel_trace.csv gpurun_out/prof_step/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/summarize_trace.py $F --skip 2 --top 80 > gpurun_out/${R}_bench_step_steady_kernel_stats.csv
python3 tools/summarize_by_grid.py $F --skip 2 --match ipsr > gpurun_out/${R}_own_kernels_by_grid.txt
rm -rf gpurun_out/prof_step
for cfg in "2:1" "4:3"; do
  c=${cfg%%:*}; p=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_layer -o layer --output-format csv -- python3 tools/profile_layer.py --cfg $c --patch $p > gpurun_out/${R}_layer_cfg${c}_p${p}.txt 2>&1
  S=$(ls gpurun_out/prof_layer/*kernel_stats.csv gpurun_out/prof_layer/*/*kernel_stats.csv 2>/dev/null | head -1)
  head -30 $S > gpurun_out/${R}_layer_cfg${c}_p${p}_kernel_stats.csv
  rm -rf gpurun_out/prof_layer
done
python3 bench.py --steps 20 > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err
python3 bench.py --steps 20 --dtype bf16 --batch 16 --no-cpu-baseline > gpurun_out/${R}_bench_bf16.json 2> gpurun_out/${R}_bench_bf16.err
# HBM traffic of the layer's kernels (two PMC passes, no other tracing) -> ${R}_layer_cfg2_hbm_traffic.csv and profiles/traffic_corr_argmax.json
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $pass -d gpurun_out/pmc_layer_$pass -o l --output-format csv -- python3 tools/profile_layer.py --cfg 2 --iters 3 > gpurun_out/pmc_layer_$pass.log 2>&1
done
FF=$(ls gpurun_out/pmc_layer_FETCH_SIZE/*counter_collection.csv gpurun_out/pmc_layer_FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
FW=$(ls gpurun_out/pmc_layer_WRITE_SIZE/*counter_collection.csv gpurun_out/pmc_layer_WRITE_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
python3 tools/collect_traffic.py $FF $FW > gpurun_out/${R}_layer_cfg2_hbm_traffic.csv && cp profiles/traffic_corr_argmax.json gpurun_out/${R}_traffic_corr_argmax.json
rm -rf gpurun_out/pmc_layer_FETCH_SIZE gpurun_out/pmc_layer_WRITE_SIZE
python3 tools/engine_map.py > gpurun_out/${R}_engine_map.txt 2> gpurun_out/${R}_engine_map.err
