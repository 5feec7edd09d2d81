cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sumcheck.py tests/test_gpu_two_round.py tests/test_gpu_mlecheck_shadow.py tests/test_gpu_multirank.py tests/test_gpu_sharded_vs_oracle.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
 echo "two-stage $(python tools/small_rounds.py 2>/dev/null | tail -1 | cut -c1-190)"
 echo "one-stage $(BN_ARM_TWO_STAGE=0 python tools/small_rounds.py 2>/dev/null | tail -1 | cut -c1-190)"
done
for i in 1 2 3; do for n in 24 25; do python bench.py --n-vars $n --steps 20 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n$n two-stage', d['ms_per_step'], d['verifier_check'])"; BN_ARM_TWO_STAGE=0 python bench.py --n-vars $n --steps 20 --warmup 3 --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n$n one-stage', d['ms_per_step'])"; done; done
