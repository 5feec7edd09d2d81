cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py tests/test_gpu_cpp_conformance.py -x -q -k "pairwise or conformance or composite" 2>&1 | tail -3
for f in 2 4; do BN_MUL9_FUSE=$f timeout 900 python -m pytest tests/test_gpu_layer.py tests/test_gpu_at_size.py -x -q -k "pairwise" 2>&1 | tail -1; done
echo "== default"; python tools/bench_pairwise.py 2>&1 | tail -4
for f in 1 2 3 4; do for m in 12 13 14 15 16; do echo "== FUSE=$f MAX_LOG2=$m $(BN_MUL9_FUSE=$f BN_PAIRTREE_MAX_LOG2=$m python tools/bench_pairwise.py 20 2>&1 | head -1)"; done; done
python tools/bench_ops.py 2>&1 | grep -i "composite\|pairwise"
tools/trace_cmd.sh r3f/trace_pair python tools/bench_pairwise.py 20; tail -6 gpurun_out/r3f/trace_pair/per_launch.jsonl
