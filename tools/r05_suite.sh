#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5l
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_group.py -q -m gpu --maxfail=6 -k "reference_piop_suite" > gpurun_out/r5l/suite.log 2>&1
tail -40 gpurun_out/r5l/suite.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_hal.py tests/test_gpu_circuits.py -q -m gpu -x > gpurun_out/r5l/hal.log 2>&1
tail -3 gpurun_out/r5l/hal.log
python tools/bench_hal.py 2>/dev/null | grep "a\*b\*c"
