#!/bin/bash
# issue priorities of the VALU-heavy side of the two large kernels (BN_FE_FP4_PRIO: the fold waves of k_foldeval_mfma_fp4;
# BN_FP4_PRIO: the staging phase of k_roundeval_fp4), levels 0 .. 3, three alternating rounds
# (profiles/r04/experiments/fe_fp4_prio.txt, fp4_prio.txt)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/prio
mkdir -p $O
cd $R
one() { # name, kernel label prefix, env..., -- bench args
  local name=$1; shift
  local kern=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[v for n,v in d['kernels'].items() if n.startswith('$kern')]
print('$name', 'ms_per_step', round(d['ms_per_step'],4), '$kern', round(k[0]['frac'],4) if k else None, d['verifier_check'], d['transcript_digest'][:12])"
}
{
for rep in 1 2 3; do
  for n in 28 22 24 25; do
    st=20; [ $n = 28 ] && st=10
    for pr in 0 1 2 3; do one "n=$n fold waves at $pr" k_foldeval_mfma BN_FE_FP4_PRIO=$pr -- --n-vars $n --steps $st --warmup 3; done
    for pr in 0 1 2 3; do one "n=$n round-0 staging at $pr" k_roundeval_fp4 BN_FP4_PRIO=$pr -- --n-vars $n --steps $st --warmup 3; done
  done
done
} > $O/times.txt 2>&1
cat $O/times.txt
