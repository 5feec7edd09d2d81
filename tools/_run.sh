cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mlecheck_shadow.py tests/test_gpu_sumcheck.py tests/test_gpu_layer.py tests/test_gpu_lazy_vs_eager.py tests/test_gpu_two_round.py -x -q 2>&1 | tail -3
BN_MLECHECK=eager tools/trace_mlecheck.sh 24 2>&1 | head -16
tools/bench_mlecheck_quick.sh 2>&1 | cut -c1-140
