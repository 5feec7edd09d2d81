cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_two_round.py tests/test_gpu_mlecheck_shadow.py tests/test_gpu_multirank.py tests/test_gpu_sharded_vs_oracle.py tests/test_gpu_lazy_vs_eager.py tests/test_gpu_at_size.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_north_star.py -x -q -k "24" 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do python tools/small_rounds.py 2>/dev/null | tail -2 | cut -c1-200; done
BNH_PROF=1 python tools/small_rounds.py 2>&1 | grep "n_vars 12" | tail -1
for i in 1 2; do python bench.py --n-vars 24 --steps 20 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n24', d['ms_per_step'], d['verifier_check'], d['transcript_digest'])"; done
