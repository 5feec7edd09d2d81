#!/bin/bash
# the round's measurement batch (run through gpurun from the repo root); results in gpurun_out/final/, copied to profiles/r06/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
# HBM traffic first: bench.py only quotes a PMC file taken on the kernel sources it runs (csrc_sha16), and the benches below
# should carry it -- so the fresh file is put where bench.py looks for it (on the box; the committed copy comes back through
# gpurun_out/final/bench_pmc.json)
BUILD=$(cat $R/tools/.build_id 2>/dev/null || python -c "import bench; print(bench.csrc_sha16())")
tools/pmc_bench.sh $BUILD 28 24 > $O/pmc_bench.log 2>&1
cp $R/gpurun_out/bench_pmc.json $O/bench_pmc.json
mkdir -p $R/profiles/r06; cp $R/gpurun_out/bench_pmc.json $R/profiles/r06/bench_pmc.json
python bench.py > $O/bench_n28.json 2> $O/bench_n28.stderr
python bench.py --n-vars 24 --steps 20 --warmup 3 > $O/bench_n24.json 2> $O/bench_n24.stderr
python bench.py --n-vars 25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n25_one_shard_of_eight.json 2>/dev/null
python bench.py --n-vars 20 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n20.json 2>/dev/null
BN_HOST_TAIL=0 python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n24_BN_HOST_TAIL_0.json 2>/dev/null
BN_HOST_TAIL=0 python bench.py --n-vars 25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n25_BN_HOST_TAIL_0.json 2>/dev/null
BN_HOST_TAIL=0 python bench.py --n-vars 20 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n20_BN_HOST_TAIL_0.json 2>/dev/null
BN_TWO_ROUND=0 python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n24_BN_TWO_ROUND_0.json 2>/dev/null
BN_TWO_ROUND=0 python bench.py --n-vars 25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n25_BN_TWO_ROUND_0.json 2>/dev/null
# ---- round 6: the keccak width (claim groups at 50 / 64 / 175 claims), the replay of config 4's HAL traffic
{ python tools/bench_piop.py claims --n-vars 22 --k 50 --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 22 --k 64 --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 22 --k 50 --kind piop --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 22 --k 100 --kind keccak --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 22 --k 40 --kind star --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 18 --k 50 --steps 10 --group 1
  python tools/bench_piop.py claims --n-vars 18 --k 100 --kind keccak --steps 10 --group 1; } > $O/claims_wide.jsonl 2> $O/claims_wide.stderr
python tools/bench_keccak_replay.py --log-perms 16 --steps 3 > $O/keccak_replay.json 2> $O/keccak_replay.stderr
python tools/bench_keccak_replay.py --log-perms 12 --steps 3 > $O/keccak_replay_2p12.json 2>/dev/null
BN_HAL_EQ_SET=0 python tools/bench_keccak_replay.py --log-perms 16 --steps 1 > $O/keccak_replay_BN_HAL_EQ_SET_0.json 2>/dev/null
python tools/bench_keccak_replay.py --table u32_add --log-rows 10 --steps 3 > $O/replay_u32_add_2e10.json 2>/dev/null
python tools/bench_keccak_replay.py --table u32_add --log-rows 20 --steps 3 > $O/replay_u32_add_2e20.json 2>/dev/null
bash tools/r06_k_sweep.sh > $O/k_sweep_balanced.txt 2>&1
bash tools/r06_k50_host.sh > $O/k50_host_phases.txt 2>&1
tools/trace_cmd.sh final/trace_claims50 python tools/bench_piop.py claims --n-vars 22 --k 50 --group 1 --steps 2 --warmup 1 > /dev/null 2>&1
cp $O/trace_claims50/kernel_stats.csv $O/claims_n22_k50_kernel_stats.csv; cp $O/trace_claims50/per_launch.jsonl $O/claims_n22_k50_per_launch.jsonl; rm -rf $O/trace_claims50
tools/trace_cmd.sh final/trace_replay python tools/bench_keccak_replay.py --log-perms 16 --steps 1 > /dev/null 2>&1
cp $O/trace_replay/kernel_stats.csv $O/keccak_replay_kernel_stats.csv; cp $O/trace_replay/per_launch.jsonl $O/keccak_replay_per_launch.jsonl; rm -rf $O/trace_replay
# ---- the PCS prover's call shape (claim groups on / off), round 5's list
{ python tools/bench_piop.py claims --n-vars 20 --k 4 --steps 10
  python tools/bench_piop.py claims --n-vars 24 --k 4 --steps 5
  python tools/bench_piop.py claims --n-vars 24 --k 4 --kind piop --steps 5
  python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5
  python tools/bench_piop.py claims --n-vars 26 --k 4 --steps 3 --group 1
  python tools/bench_piop.py claims --n-vars 26 --k 4 --kind bipartite --steps 3 --group 1
  BN_GROUP_CHAIN_MIN_LOG2=63 python tools/bench_piop.py claims --n-vars 26 --k 4 --kind bipartite --steps 3 --group 1
  python tools/bench_piop.py claims --n-vars 26 --k 4 --kind piop --steps 3 --group 1
  BN_GROUP_CHAIN_MIN_LOG2=63 python tools/bench_piop.py claims --n-vars 26 --k 4 --kind piop --steps 3 --group 1
  BN_GROUP_CHAIN_MIN_LOG2=0 python tools/bench_piop.py claims --n-vars 20 --k 4 --kind bipartite --steps 10 --group 1
  python tools/bench_piop.py claims --n-vars 20 --k 4 --kind bipartite --steps 10 --group 1
  python tools/bench_piop.py claims --n-vars 24 --k 8 --steps 3 --group 1
  python tools/bench_piop.py claims --n-vars 16 --k 4 --steps 20
  python tools/bench_piop.py claims --n-vars 12 --k 8 --steps 20
  python tools/bench_piop.py piop --n 20 --steps 5
  python tools/bench_piop.py piop --n 16 --steps 10
  python tools/bench_piop.py piop --n 12 --steps 10
  BN_GROUP_HT_MAX_LOG2=0 python tools/bench_piop.py piop --n 12 --steps 10 --group 1
  BN_GROUP_SPEC=0 python tools/bench_piop.py piop --n 20 --steps 5 --group 1; } > $O/piop.jsonl 2> $O/piop.stderr
tools/trace_cmd.sh final/trace_claims26 python tools/bench_piop.py claims --n-vars 26 --k 4 --group 1 --steps 2 --warmup 1 > /dev/null 2>&1
cp $O/trace_claims26/kernel_stats.csv $O/claims_n26_k4_kernel_stats.csv; cp $O/trace_claims26/per_launch.jsonl $O/claims_n26_k4_per_launch.jsonl; rm -rf $O/trace_claims26
tools/trace_cmd.sh final/trace_piop20 python tools/bench_piop.py piop --n 20 --group 1 --steps 2 --warmup 1 > /dev/null 2>&1
cp $O/trace_piop20/kernel_stats.csv $O/piop_n20_kernel_stats.csv; cp $O/trace_piop20/per_launch.jsonl $O/piop_n20_per_launch.jsonl; rm -rf $O/trace_piop20
python tools/bench_ops.py > $O/ops.jsonl 2>&1
tools/bench_mlecheck_quick.sh > $O/mlecheck_prover.jsonl 2>&1
BN_MLECHECK_SHADOW=0 tools/bench_mlecheck_quick.sh > $O/mlecheck_prover_BN_MLECHECK_SHADOW_0.jsonl 2>&1
python tools/profile_ntt.py --reps 3 > $O/ntt_2p24_b32.txt 2>&1
python tools/bench_fri_commit.py > $O/fri_commit.jsonl 2>&1
python tools/bench_merkle.py --log-n 24 --batches 4 16 64 > $O/merkle.jsonl 2>&1
python tools/small_rounds.py > $O/small_rounds.jsonl 2>&1
BN_TWO_ROUND=0 python tools/small_rounds.py > $O/small_rounds_BN_TWO_ROUND_0.jsonl 2>&1
BN_HOST_TAIL=0 python tools/small_rounds.py > $O/small_rounds_BN_HOST_TAIL_0.jsonl 2>&1
python tools/bench_pairwise.py > $O/pairwise.jsonl 2>&1
python tools/bench_hal.py > $O/hal.jsonl 2>&1
# the old HAL's degree-3 requests: coefficient form beside the general code (BN_HAL_COEF=0), one / three domain points, with an indicator
{ for eq in 0 1; do for p in 1 3; do
    python tools/bench_hal_cubic.py --eq $eq --points $p
    BN_HAL_COEF=0 python tools/bench_hal_cubic.py --eq $eq --points $p
  done; done; } > $O/hal_cubic.jsonl 2> $O/hal_cubic.stderr
{ for n in 20 16; do python tools/bench_zerocheck_cubic.py --n-vars $n; BN_HAL_COEF=0 python tools/bench_zerocheck_cubic.py --n-vars $n; done; } > $O/zerocheck_cubic.jsonl 2>/dev/null
tools/trace_cmd.sh final/trace_cubic python tools/bench_hal_cubic.py --n-vars 24 --reps 2 > /dev/null 2>&1
cp $O/trace_cubic/per_launch.jsonl $O/hal_cubic_n24_per_launch.jsonl; rm -rf $O/trace_cubic
tools/trace_bench.sh final/trace_n28 --n-vars 28 --steps 3 --warmup 1 --no-cpu-baseline --no-claim-groups > /dev/null 2>&1
tools/trace_bench.sh final/trace_n24 --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline --no-claim-groups > /dev/null 2>&1
for t in trace_n28 trace_n24; do cp $O/$t/kernel_stats.csv $O/bench_${t#trace_}_kernel_stats.csv; cp $O/$t/per_launch.jsonl $O/per_launch_${t#trace_}.jsonl; cp $O/$t/bench_line.json $O/bench_${t#trace_}_under_rocprof.json; done
rm -rf $O/trace_n28 $O/trace_n24
# config 5's workload and the exchanges, all ranks on this one device (diagnostic: eight processes share one GPU); the bare
# `--gpus N` form (no launcher: bench.py spawns its own ranks)
for W in 2 4 8; do
  BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 300 python bench.py --gpus $W --n-vars 13 --steps 50 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | grep '^{' > $O/bench_${W}_ranks_on_one_gpu_n13.json
done
BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_8_ranks_on_one_gpu_n28.json
tail -c 700 $O/bench_n28.json
