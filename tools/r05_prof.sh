#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5e
export TMPDIR=/tmp
{
BNH_PROF=1 python tools/bench_piop.py piop --n 20 --group 1 --steps 2 --warmup 1
BNH_PROF=1 python tools/bench_piop.py piop --n 12 --group 1 --steps 2 --warmup 1
BNH_PROF=1 python tools/bench_piop.py piop --n 12 --group 0 --steps 2 --warmup 1
timeout 600 python -m pytest tests/test_gpu_group_fuzz.py -q -m gpu -x 2>&1 | tail -3
} > gpurun_out/r5e/prof.log 2>&1
grep -v "^{" gpurun_out/r5e/prof.log | cut -c1-400
