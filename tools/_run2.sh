cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_at_size.py tests/test_gpu_sumcheck.py tests/test_gpu_north_star.py -m gpu -x -q 2>&1 | tail -3
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[v for n,v in d['kernels'].items() if n.startswith('k_roundeval_fp4')]
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'round0', round(k[0]['frac'],4) if k else None, d['verifier_check'], d['transcript_digest'][:12])"
}
for rep in 1 2 3; do
  for n in 28 25 24 22; do
    st=20; [ $n = 28 ] && st=10
    one "n=$n ws2" X=1 -- --n-vars $n --steps $st --warmup 3
    one "n=$n shipped" BN_FP4_WS=0 -- --n-vars $n --steps $st --warmup 3
  done
done
python tools/run_ip_power.py 27 | grep -E "random|zeros"
BN_FP4_WS=0 python tools/run_ip_power.py 27 | grep -E "random|zeros"
