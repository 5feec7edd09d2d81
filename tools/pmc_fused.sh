#!/bin/bash
# counters-only rocprofv3 passes over tools/run_fused_once.py (fused launches of ONE size): instruction mix, issue-busy cycles
# and the matrix pipe's busy cycles of k_foldeval_mfma_fp4; summary in gpurun_out/pmc_fused/summary.json
LOG=${1:-27}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_fused
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && BN_ARM=0 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/run_fused_once.py $LOG 3 > $OUT/p$i.log 2>&1)
done
python3 $R/tools/pmc_summary.py $OUT/summary_all.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 > /dev/null
rm -rf $OUT/p[0-9]
python3 - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/summary_all.json"))
keep = {k: v for k, v in d.items() if "foldeval_mfma" in k or "roundeval" in k}
json.dump(keep, open(sys.argv[1] + "/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(keep, indent=1, sort_keys=True))
PY
(cd $R && BN_ARM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/run_fused_once.py $LOG 3 > $OUT/trace.log 2>&1)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
grep -i "foldeval\|roundeval" $OUT/kernel_stats.csv | cut -c1-200
