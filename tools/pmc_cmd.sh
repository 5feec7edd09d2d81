#!/bin/bash
# counters-only rocprofv3 passes over an arbitrary command; summary per kernel in gpurun_out/<name>/summary.json
# usage: tools/pmc_cmd.sh <name> <command...>
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1)
done
python3 $R/tools/pmc_summary.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 > /dev/null
rm -rf $OUT/p[0-9]
python3 -c "
import json,sys
d=json.load(open('$OUT/summary.json'))
for k,v in d.items():
    if 'fri' in k or 'extrapolate' in k: print(k[:60], json.dumps(v))
"
