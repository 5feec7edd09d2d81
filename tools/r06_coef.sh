#!/bin/bash
# the coefficient-form path of the old HAL's round evaluation: parity, the tests around it, tools/bench_hal.py on and off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/coef; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_hal_coef.py tests/test_gpu_hal.py tests/test_gpu_hal_wide.py tests/test_gpu_zerocheck.py tests/test_gpu_circuits.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python tools/bench_hal.py > $O/hal.jsonl 2> $O/hal.err; tail -3 $O/hal.err
BN_HAL_COEF=0 timeout 300 python tools/bench_hal.py > $O/hal_BN_HAL_COEF_0.jsonl 2>> $O/hal.err
grep -h "a\*b\*c" $O/hal.jsonl $O/hal_BN_HAL_COEF_0.jsonl
