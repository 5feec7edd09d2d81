#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/coef3; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_hal_coef.py tests/test_gpu_zerocheck.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 200 python tools/bench_hal_cubic.py --eq 0 --points 1 >> $O/cubic.jsonl 2>> $O/err.log
timeout 200 python tools/bench_hal_cubic.py --eq 1 --points 1 >> $O/cubic.jsonl 2>> $O/err.log
cat $O/cubic.jsonl
bash tools/trace_cmd.sh coef3/trace python tools/bench_hal_cubic.py --n-vars 24 --reps 2
tail -8 $O/trace/per_launch.jsonl
timeout 200 python tools/bench_ops.py 2>/dev/null | grep -i "composite" 
