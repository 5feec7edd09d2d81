cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_mlecheck_shadow.py tests/test_gpu_two_round.py tests/test_gpu_lazy_vs_eager.py -x -q 2>&1 | tail -2
for i in 1 2; do tools/bench_mlecheck_quick.sh 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['op'][-20:], d['prover'], d['ms'])"; done
