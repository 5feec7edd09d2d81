cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_at_size.py tests/test_gpu_sumcheck.py tests/test_gpu_north_star.py tests/test_gpu_lazy_vs_eager.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --n-vars 28 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=28', round(d['ms_per_step'],4), {k[:16]:v['frac'] for k,v in d['kernels'].items()})"; done
for n in 24 25; do python bench.py --n-vars $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n=$n', round(d['ms_per_step'],4), {k[:16]:v['frac'] for k,v in d['kernels'].items()})"; done
