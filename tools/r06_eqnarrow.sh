#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eqnarrow; rm -rf $O; mkdir -p $O; cd $R
for n in 24 22 20 18 16 13; do for thr in 0 99; do
echo "n=$n BN_HAL_EQ_SET_NARROW_MIN_LOG2=$thr" >> $O/sweep.txt
BN_HAL_EQ_SET_NARROW_MIN_LOG2=$thr timeout 200 python tools/bench_hal.py --n-vars $n --n-vars-general 12 2>> $O/err.log | grep -F "(a*b + c) * eq" >> $O/sweep.txt
done; done
cat $O/sweep.txt
timeout 600 python -m pytest tests/test_gpu_hal.py tests/test_gpu_zerocheck.py tests/test_gpu_sumcheck.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
BN_HAL_EQ_SET_NARROW_MIN_LOG2=0 timeout 600 python -m pytest tests/test_gpu_hal.py tests/test_gpu_zerocheck.py -x -q > $O/pytest_thr0.log 2>&1; tail -3 $O/pytest_thr0.log
