cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prio
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[v for n,v in d['kernels'].items() if n.startswith('k_roundeval_fp4')]
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'fused', round(k[0]['frac'],4) if k else None, d['verifier_check'], d['transcript_digest'][:12])"
}
{
for rep in 1 2 3; do
  for pr in 0 1 2 3; do one "n=28 round-0 prio $pr" BN_FP4_PRIO=$pr -- --n-vars 28 --steps 10 --warmup 3; done
  for n in 22 24 25; do
    for pr in 0 1 2 3; do one "n=$n round-0 prio $pr" BN_FP4_PRIO=$pr -- --n-vars $n --steps 20 --warmup 3; done
  done
done
} > gpurun_out/prio/times.txt 2>&1
cat gpurun_out/prio/times.txt
