cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py tests/test_gpu_fri.py tests/test_gpu_cpp_conformance.py -x -q -k "fri or conformance" 2>&1 | tail -3
python tools/bench_ops.py 2>&1 | tee $O/ops.jsonl | grep -i "fri\|pairwise\|NTT\|composite"
BN_FRI_MULTI=0 python tools/bench_ops.py 2>&1 | grep -i "fri"
python tools/bench_fri_commit.py 2>&1 | tail -3
