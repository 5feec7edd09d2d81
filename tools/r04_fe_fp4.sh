#!/bin/bash
# the wave-specialised FP4 form of the fused fold + evaluation kernel (kernels_foldeval_fp4.hip) against the int8 form
# (kernels_foldeval_mfma.hip, BN_FE_FP4=0): step times alternating, then the MLE-check prover (the scaled folds)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fe_fp4
mkdir -p $O
cd $R
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[v for n,v in d['kernels'].items() if n.startswith('k_foldeval_mfma')]
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'fused', round(k[0]['frac'],4) if k else None, d['verifier_check'], d['transcript_digest'][:12])"
}
{
for rep in 1 2; do
  one "n=28 fp4" X=1 -- --n-vars 28 --steps 10 --warmup 3
  one "n=28 int8" BN_FE_FP4=0 -- --n-vars 28 --steps 10 --warmup 3
  for n in 20 22 24 25; do
    one "n=$n fp4" X=1 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n int8" BN_FE_FP4=0 -- --n-vars $n --steps 20 --warmup 3
  done
done
for v in 1 0 1 0; do BN_FE_FP4=$v tools/bench_mlecheck_quick.sh 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    m = re.search(r'n_vars=(\d+).*\"prover\": \"(\w+)\", \"ms\": ([\d.]+)', l)
    if m: print('MLE-check prove n_vars=%s m=2 %-8s BN_FE_FP4=$v: %s ms' % (m.group(1), m.group(2), m.group(3)))"; done
} > $O/step_times.txt 2>&1
cat $O/step_times.txt
