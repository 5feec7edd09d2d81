#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lone; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_hal.py tests/test_gpu_hal_coef.py tests/test_gpu_hal_wide.py tests/test_gpu_zerocheck.py tests/test_gpu_circuits.py tests/test_gpu_cpp_conformance.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/bench_hal.py > $O/hal.jsonl 2>> $O/err.log; cut -c1-200 $O/hal.jsonl
