cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py tests/test_gpu_fri.py tests/test_gpu_cpp_conformance.py -x -q -k "fri or conformance" 2>&1 | tail -2
BN_FRI_NTT_C3=1 timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py tests/test_gpu_fri.py -x -q -k "fri" 2>&1 | tail -2
python tools/bench_ops.py 2>&1 | grep -i "fri"
BN_FRI_NTT_C3=1 python tools/bench_ops.py 2>&1 | grep -i "fri"
tools/trace_cmd.sh r3f/trace_fri python tools/run_fri_only.py; tail -3 gpurun_out/r3f/trace_fri/per_launch.jsonl
BN_FRI_NTT_C3=1 tools/trace_cmd.sh r3f/trace_fri3 python tools/run_fri_only.py; tail -3 gpurun_out/r3f/trace_fri3/per_launch.jsonl
python tools/bench_fri_commit.py 2>&1 | tail -1 | cut -c1-300
