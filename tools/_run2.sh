cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mlecheck_shadow.py tests/test_gpu_sumcheck.py tests/test_gpu_north_star.py tests/test_gpu_layer.py -m gpu -x -q 2>&1 | tail -3
bash tools/r04_fe_fp4.sh
