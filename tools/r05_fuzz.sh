#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5d
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_group_fuzz.py -q -m gpu --maxfail=6 > gpurun_out/r5d/fuzz.log 2>&1
tail -60 gpurun_out/r5d/fuzz.log | cut -c1-300
