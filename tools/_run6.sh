cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; rm -rf $O; mkdir -p $O
tools/trace_cmd.sh r3f/trace_fri python tools/run_fri_only.py; tail -12 gpurun_out/r3f/trace_fri/per_launch.jsonl
BN_FRI_MULTI=0 tools/trace_cmd.sh r3f/trace_fri_old python tools/run_fri_only.py; tail -8 gpurun_out/r3f/trace_fri_old/per_launch.jsonl
tools/trace_bench.sh r3f/trace_n28 --n-vars 28 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
tools/trace_bench.sh r3f/trace_n24 --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
tools/trace_bench.sh r3f/trace_n25 --n-vars 25 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
head -12 gpurun_out/r3f/trace_n28/kernel_stats.csv | cut -c1-200
tail -45 gpurun_out/r3f/trace_n24/per_launch.jsonl
