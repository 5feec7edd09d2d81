cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py tests/test_gpu_fri.py tests/test_gpu_cpp_conformance.py -x -q -k "fri or conformance" 2>&1 | tail -2
for i in 1 2 3; do python tools/bench_ops.py 2>&1 | grep -i "fri" | cut -c1-140; done
tools/trace_cmd.sh r3f/trace_fri python tools/run_fri_only.py; tail -3 gpurun_out/r3f/trace_fri/per_launch.jsonl
python tools/bench_fri_commit.py 2>&1 | tail -1 | cut -c150-300
