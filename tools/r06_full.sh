#!/bin/bash
# the whole GPU suite + smoke, then the round's measurement batch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/verify; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -3 $O/pytest.log; tail -1 $O/smoke.log
bash tools/final_measure.sh
