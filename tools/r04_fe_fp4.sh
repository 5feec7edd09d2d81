#!/bin/bash
# the wave-specialised FP4 form of the fused fold + evaluation kernel (kernels_foldeval_fp4.hip; BN_FE_FP4=0 = off,
# BN_FE_FP4_MIN_LOG2 = elements per array from which it takes the launch): parity tests, then step times alternating
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fe_fp4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_north_star.py tests/test_gpu_sumcheck.py tests/test_gpu_lazy_vs_eager.py tests/test_gpu_at_size.py tests/test_gpu_layer.py -m gpu -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'ms_per_step', round(d['ms_per_step'],4), 'roofline', round(d['roofline']['frac'],4), d['verifier_check'], d['transcript_digest'][:12], json.dumps(d.get('kernels'))[:600])"
}
{
for rep in 1 2; do
  one "n=28 ws" X=1 -- --n-vars 28 --steps 10 --warmup 3
  one "n=28 r0 old" BN_FP4_WS=0 -- --n-vars 28 --steps 10 --warmup 3
  one "n=28 both old" BN_FP4_WS=0 BN_FE_FP4=0 -- --n-vars 28 --steps 10 --warmup 3
  for n in 22 24 25; do
    one "n=$n ws" X=1 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n r0 old" BN_FP4_WS=0 -- --n-vars $n --steps 20 --warmup 3
    one "n=$n both old" BN_FP4_WS=0 BN_FE_FP4=0 -- --n-vars $n --steps 20 --warmup 3
  done
done
} > $O/step_times.txt 2>&1
cut -c1-330 $O/step_times.txt
