#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5i
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_hal.py tests/test_gpu_circuits.py tests/test_gpu_cpp_conformance.py -q -m gpu --maxfail=8 > gpurun_out/r5i/tests.log 2>&1
tail -25 gpurun_out/r5i/tests.log | cut -c1-250
python tools/bench_hal.py > gpurun_out/r5i/hal.jsonl 2> gpurun_out/r5i/hal.err
cat gpurun_out/r5i/hal.jsonl | cut -c1-400
tail -3 gpurun_out/r5i/hal.err
