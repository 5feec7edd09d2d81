cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_at_size.py -m gpu -x -q -k "odd_tile_counts or compiled_sumcheck_plan" 2>&1 | tail -5
