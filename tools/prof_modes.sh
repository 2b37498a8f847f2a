export TMPDIR=/tmp IPSR_BENCH_STEP_ONLY=1
for mode in "x6:--conv-math bf16x6" "x3:--conv-math bf16x3" "bf16:--dtype bf16 --batch 16"; do
  tag=${mode%%:*}; args=${mode#*:}
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p --output-format csv -- python bench.py --steps 10 --warmup 4 --no-cpu-baseline $args > gpurun_out/r3_prof_$tag.json 2> gpurun_out/r3_prof_$tag.err
  F=$(ls gpurun_out/prof_$tag/*kernel_stats.csv gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  head -70 $F > gpurun_out/r3_prof_${tag}_stats.csv
  rm -rf gpurun_out/prof_$tag
done
