#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_group.py -q -m gpu -x -k "not 20- and not 22- and not 17" > gpurun_out/r5c/group_small.log 2>&1
tail -3 gpurun_out/r5c/group_small.log
{
python tools/bench_piop.py claims --n-vars 20 --k 4 --steps 10 --group 1
python tools/bench_piop.py claims --n-vars 24 --k 4 --steps 5 --group 1
python tools/bench_piop.py claims --n-vars 24 --k 4 --kind piop --steps 5 --group 1
python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5 --group 1
python tools/bench_piop.py claims --n-vars 26 --k 4 --steps 3 --group 1
python tools/bench_piop.py claims --n-vars 26 --k 1 --steps 3 --group 1
python tools/bench_piop.py claims --n-vars 16 --k 4 --steps 20 --group 1
python tools/bench_piop.py piop --n 20 --steps 5 --group 1
} > gpurun_out/r5c/bench.log 2>&1
grep -v "^\[" gpurun_out/r5c/bench.log | cut -c1-700
