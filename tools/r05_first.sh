#!/bin/bash
# round 5, first GPU batch: the new claim-group tests, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_circuits.py -q -m gpu --maxfail=12 -k "group or fresh_context" > gpurun_out/r5a/group.log 2>&1
echo "group rc=$?" >> gpurun_out/r5a/group.log
tail -40 gpurun_out/r5a/group.log
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_group.py > gpurun_out/r5a/full.log 2>&1
echo "full rc=$?" >> gpurun_out/r5a/full.log
tail -15 gpurun_out/r5a/full.log
