#!/bin/bash
# the round's measurement batch (run through gpurun from the repo root); results in gpurun_out/final/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_n28.json 2> $O/bench_n28.stderr
python bench.py --n-vars 24 --steps 5 --warmup 2 > $O/bench_n24.json 2> $O/bench_n24.stderr
BN_EVAL=valu python bench.py --n-vars 28 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_n28_valu_kernels_BN_EVAL_valu.json 2>/dev/null
python tools/bench_ops.py > $O/ops.jsonl 2>&1
tools/bench_mlecheck_quick.sh > $O/mlecheck_prover.jsonl 2>&1
python tools/profile_ntt.py --reps 3 > $O/ntt_2p24_b32.txt 2>&1
python tools/bench_fri_commit.py > $O/fri_commit.jsonl 2>&1
python tools/roundeval_rate.py --n-vars 24 26 27 > $O/roundeval_rate.jsonl 2>&1
python tools/small_rounds.py > $O/small_rounds.jsonl 2>&1
BN_ARM=0 python tools/small_rounds.py > $O/small_rounds_BN_ARM_0.jsonl 2>&1
BN_ARM=0 python bench.py --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n24_BN_ARM_0.json 2>/dev/null
python bench.py --n-vars 20 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n20.json 2>/dev/null
tools/small_round_phases > $O/small_round_phases.txt 2>&1
tools/signal_latency > $O/signal_latency.txt 2>&1
python tools/bench_hal.py > $O/hal.jsonl 2>&1
tools/trace_bench.sh trace_n28_final --n-vars 28 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
tools/trace_bench.sh trace_n24_final --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
tools/trace_mlecheck.sh 24 > $O/mlecheck_timeline_n24.txt 2>&1
for t in trace_n28_final trace_n24_final; do cp $R/gpurun_out/$t/kernel_stats.csv $O/${t}_kernel_stats.csv; cp $R/gpurun_out/$t/per_launch.jsonl $O/${t}_per_launch.jsonl; cp $R/gpurun_out/$t/bench_line.json $O/${t}_bench_line.json; done
tail -c 600 $O/bench_n28.json
