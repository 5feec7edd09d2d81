#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_group.py -q -m gpu --maxfail=8 -k "not 20- and not 22- and not 17" > gpurun_out/r5f/group_small.log 2>&1
tail -30 gpurun_out/r5f/group_small.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_group_fuzz.py -q -m gpu --maxfail=6 > gpurun_out/r5f/fuzz.log 2>&1
tail -30 gpurun_out/r5f/fuzz.log | cut -c1-250
{
BNH_PROF=1 python tools/bench_piop.py piop --n 20 --group 1 --steps 3 --warmup 1
BNH_PROF=1 python tools/bench_piop.py piop --n 12 --group 1 --steps 3 --warmup 1
python tools/bench_piop.py claims --n-vars 16 --k 4 --steps 20 --group 1
python tools/bench_piop.py claims --n-vars 12 --k 8 --steps 20 --group 1
python tools/bench_piop.py claims --n-vars 20 --k 4 --steps 20 --group 1
} > gpurun_out/r5f/bench.log 2>&1
grep -v "^\[" gpurun_out/r5f/bench.log | cut -c1-600
grep "bnh prof" gpurun_out/r5f/bench.log | tail -4
