#!/bin/bash
# the GPU suite twice more on one box (flake hunt)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/flake; rm -rf $O; mkdir -p $O; cd $R
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_$i.log 2>&1; echo "pytest rc=$?" >> $O/pytest_$i.log
  tail -2 $O/pytest_$i.log
done
