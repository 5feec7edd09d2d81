#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/groestl; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri.py tests/test_gpu_cpp_conformance.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/bench_merkle.py --log-n 24 --batches 4 16 64 >> $O/merkle.txt 2>> $O/err.log
timeout 300 python tools/bench_fri_commit.py >> $O/merkle.txt 2>> $O/err.log
cut -c1-330 $O/merkle.txt
