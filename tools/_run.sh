cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py tests/test_gpu_cpp_conformance.py -x -q -k "fold or conformance" 2>&1 | tail -2
python tools/bench_ops.py 2>&1 | grep -i "fold_r"
BN_LINMAP=0 python tools/bench_ops.py 2>&1 | grep -i "fold_r"
