#!/bin/bash
# the round's measurement batch (run through gpurun from the repo root); results in gpurun_out/final/, copied to profiles/r04/
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
mkdir -p $R/profiles/r04; cp $R/gpurun_out/bench_pmc.json $R/profiles/r04/bench_pmc.json
python bench.py > $O/bench_n28.json 2> $O/bench_n28.stderr
python bench.py --n-vars 24 --steps 20 --warmup 3 > $O/bench_n24.json 2> $O/bench_n24.stderr
python bench.py --n-vars 25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n25_one_shard_of_eight.json 2>/dev/null
python bench.py --n-vars 20 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n20.json 2>/dev/null
BN_HOST_TAIL=0 python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n24_BN_HOST_TAIL_0.json 2>/dev/null
BN_HOST_TAIL=0 python bench.py --n-vars 25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n25_BN_HOST_TAIL_0.json 2>/dev/null
BN_HOST_TAIL=0 python bench.py --n-vars 20 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n20_BN_HOST_TAIL_0.json 2>/dev/null
BN_TWO_ROUND=0 python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n24_BN_TWO_ROUND_0.json 2>/dev/null
BN_TWO_ROUND=0 python bench.py --n-vars 25 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n25_BN_TWO_ROUND_0.json 2>/dev/null
python tools/bench_ops.py > $O/ops.jsonl 2>&1
tools/bench_mlecheck_quick.sh > $O/mlecheck_prover.jsonl 2>&1
BN_MLECHECK_SHADOW=0 tools/bench_mlecheck_quick.sh > $O/mlecheck_prover_BN_MLECHECK_SHADOW_0.jsonl 2>&1
python tools/profile_ntt.py --reps 3 > $O/ntt_2p24_b32.txt 2>&1
python tools/bench_fri_commit.py > $O/fri_commit.jsonl 2>&1
python tools/small_rounds.py > $O/small_rounds.jsonl 2>&1
BN_TWO_ROUND=0 python tools/small_rounds.py > $O/small_rounds_BN_TWO_ROUND_0.jsonl 2>&1
BN_HOST_TAIL=0 python tools/small_rounds.py > $O/small_rounds_BN_HOST_TAIL_0.jsonl 2>&1
tools/mfma_round_phases > $O/mfma_round_phases.txt 2>&1
tools/r04_fe_variants.sh > /dev/null 2>&1; cp $R/gpurun_out/fe_variants/times.txt $O/fe_variants.txt
python tools/bench_pairwise.py > $O/pairwise.jsonl 2>&1
BN_PAIRTREE_MAX_LOG2=0 python tools/bench_pairwise.py 20 > $O/pairwise_BN_PAIRTREE_MAX_LOG2_0.jsonl 2>&1
tools/trace_cmd.sh final/trace_pair python tools/bench_pairwise.py 20 > /dev/null 2>&1; tail -8 $O/trace_pair/per_launch.jsonl > $O/pairwise_per_launch.jsonl; rm -rf $O/trace_pair
tools/trace_cmd.sh final/trace_fri python tools/run_fri_only.py > /dev/null 2>&1; tail -3 $O/trace_fri/per_launch.jsonl > $O/fri_fold_per_launch.jsonl; rm -rf $O/trace_fri
{ for rep in 1 2 3; do for n in 20 24 25; do for HT in 1 0; do BN_HOST_TAIL=$HT python bench.py --n-vars $n --steps 20 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=$n BN_HOST_TAIL=$HT ms_per_step', round(d['ms_per_step'],4), d['verifier_check'], d['transcript_digest'])"; done; done; done
  echo "== BN_ARM_MAX_LOG2=21 (round 3's arming limit)"; for n in 24 25; do BN_ARM_MAX_LOG2=21 python bench.py --n-vars $n --steps 20 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=$n ms_per_step', round(d['ms_per_step'],4), d['verifier_check'], d['transcript_digest'])"; done
  echo "== BN_TWO_ROUND=0"; for n in 20 24 25; do BN_TWO_ROUND=0 python bench.py --n-vars $n --steps 20 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=$n off ms_per_step', d['ms_per_step'], d['verifier_check'], d['transcript_digest'])"; done; } > $O/two_round_step_times.txt 2>&1
tools/two_round_phases > $O/two_round_phases.txt 2>&1
tools/small_round_phases > $O/small_round_phases.txt 2>&1
python tools/bench_hal.py > $O/hal.jsonl 2>&1
# round 4, second half: A/B of the kernel changes on this box (NTT register passes, two batches per rebuild, folds on the matrix cores)
tools/r04_ab2.sh > /dev/null 2>&1; cp $R/gpurun_out/ab2/times.txt $O/ab_ntt_regpass_mul9_dual.txt
{ for v in 1 0 1 0; do echo "BN_FOLD_MFMA=$v"; BN_FOLD_MFMA=$v python tools/bench_ops.py 2>&1 | grep -E "fold_"; done; } > $O/ab_fold_mfma.txt 2>&1
tools/trace_cmd.sh final/trace_fold python tools/run_fold_only.py > /dev/null 2>&1; tail -12 $O/trace_fold/per_launch.jsonl > $O/fold_per_launch.jsonl; rm -rf $O/trace_fold
tools/r04_nt.sh > /dev/null 2>&1; cp $R/gpurun_out/nt/step_times.txt $O/ab_nt_step_times.txt
# round 4, last step: the fused kernel's two forms
tools/r04_fe_fp4.sh > /dev/null 2>&1; cp $R/gpurun_out/fe_fp4/step_times.txt $O/ab_fe_fp4.txt
# counters of the two large kernels at one size each (instruction mix, pipe-busy cycles)
tools/pmc_fused.sh 27 > /dev/null 2>&1; cp $R/gpurun_out/pmc_fused/summary.json $O/fused_fp4_pmc_2p27.json; cp $R/gpurun_out/pmc_fused/kernel_stats.csv $O/fused_fp4_2p27_kernel_stats.csv
tools/pmc_round0.sh 27 > /dev/null 2>&1; cp $R/gpurun_out/pmc_round0/summary.json $O/round0_fp4_pmc_2p27.json; cp $R/gpurun_out/pmc_round0/kernel_stats.csv $O/round0_fp4_2p27_kernel_stats.csv
tools/trace_bench.sh final/trace_n28 --n-vars 28 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
tools/trace_bench.sh final/trace_n24 --n-vars 24 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for t in trace_n28 trace_n24; do cp $O/$t/kernel_stats.csv $O/bench_${t#trace_}_kernel_stats.csv; cp $O/$t/per_launch.jsonl $O/per_launch_${t#trace_}.jsonl; cp $O/$t/bench_line.json $O/bench_${t#trace_}_under_rocprof.json; done
rm -rf $O/trace_n28 $O/trace_n24
BN_MLECHECK=eager tools/trace_mlecheck.sh 24 > $O/mlecheck_literal_timeline_n24.txt 2>&1
# config 5's workload and the exchanges, all ranks on this one device (diagnostic: eight processes share one GPU)
for W in 2 4 8; do
  BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2971$W bench.py --gpus $W --n-vars 13 --steps 50 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | grep '^{' > $O/bench_${W}_ranks_on_one_gpu_n13.json
done
BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29719 bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_8_ranks_on_one_gpu_n28.json
tail -c 700 $O/bench_n28.json
