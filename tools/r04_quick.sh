#!/bin/bash
# quick measurement batch of round 4 (run through gpurun from the repo root): step times of the single-GPU prover at the
# sizes that decide the 8-GPU budget, with the host tail on and off; results in gpurun_out/quick/
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/quick
mkdir -p $O
cd $R
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'ms_per_step', round(d['ms_per_step'],4), d['verifier_check'], d['transcript_digest'])"
}
{
for rep in 1 2; do
  for n in 20 24 25; do
    one "n=$n default" X=1 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n BN_HOST_TAIL=0" BN_HOST_TAIL=0 -- --n-vars $n --steps 20 --warmup 3
  done
done
one "n=28 default" X=1 -- --n-vars 28 --steps 10 --warmup 3
} > $O/step_times.txt 2>&1
BNH_PROF=1 python bench.py --n-vars 24 --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2> $O/bnh_prof_n24.txt > /dev/null
python tools/small_rounds.py > $O/small_rounds.jsonl 2>&1
BN_HOST_TAIL=0 python tools/small_rounds.py > $O/small_rounds_BN_HOST_TAIL_0.jsonl 2>&1
cat $O/step_times.txt
tail -3 $O/bnh_prof_n24.txt
tail -4 $O/small_rounds.jsonl
# the exchanges with several ranks on this one device (diagnostic): the transcript must be the single-GPU one
for W in 2 8; do
  BN_ALL_ON_GPU0=1 BN_PG_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2972$W bench.py --gpus $W --n-vars 13 --steps 50 --warmup 5 --no-cpu-baseline --no-prof 2>/dev/null | grep '^{' > $O/bench_${W}_ranks_on_one_gpu_n13.json
  python -c "import json; d=json.load(open('$O/bench_${W}_ranks_on_one_gpu_n13.json')); print('$W ranks n=13:', round(d['ms_per_step'],4), d['verifier_check'], d['transcript_digest'], json.dumps(d['exchanges']))"
done
