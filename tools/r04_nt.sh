#!/bin/bash
# non-temporal accesses in the two large kernels (BN_FE_NT_MIN_LOG2 / BN_FP4_NT_MIN_LOG2; 64 = off): step times and the bench's
# own roofline figures, alternating
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/nt
mkdir -p $O
cd $R
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'roofline', round(d['roofline']['frac'],4), d['verifier_check'], d['transcript_digest'][:12], json.dumps(d.get('kernels'))[:400])"
}
{
for rep in 1 2; do
  one "n=28 default" X=1 -- --n-vars 28 --steps 10 --warmup 3
  one "n=28 FE off" BN_FE_NT_MIN_LOG2=64 -- --n-vars 28 --steps 10 --warmup 3
  one "n=28 FP4 off" BN_FP4_NT_MIN_LOG2=64 -- --n-vars 28 --steps 10 --warmup 3
  one "n=28 both off" BN_FE_NT_MIN_LOG2=64 BN_FP4_NT_MIN_LOG2=64 -- --n-vars 28 --steps 10 --warmup 3
  for n in 24 25; do
    one "n=$n default" X=1 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n FE>=23" BN_FE_NT_MIN_LOG2=23 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n FE>=22 FP4>=22" BN_FE_NT_MIN_LOG2=22 BN_FP4_NT_MIN_LOG2=22 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n both off" BN_FE_NT_MIN_LOG2=64 BN_FP4_NT_MIN_LOG2=64 -- --n-vars $n --steps 20 --warmup 3
  done
done
} > $O/step_times.txt 2>&1
cat $O/step_times.txt
