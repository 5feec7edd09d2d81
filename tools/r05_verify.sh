#!/bin/bash
# full GPU suite + smoke + the old-HAL bench on the current build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/verify; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python tools/bench_hal.py > $O/hal.jsonl 2>&1
tail -3 $O/pytest.log; tail -1 $O/smoke.log; cat $O/hal.jsonl | cut -c1-300
