#!/bin/bash
# counters-only rocprofv3 passes over tools/run_ip_power.py (inner_product of two 2^LOG-element vectors = k_roundeval_fp4 alone,
# on zero / sparse / random inputs): instruction mix, issue-busy cycles, matrix pipe; summary in gpurun_out/pmc_round0/summary.json
LOG=${1:-27}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_round0
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd $R && rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/run_ip_power.py $LOG > $OUT/p$i.log 2>&1)
done
python3 $R/tools/pmc_summary.py $OUT/summary_all.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 > /dev/null
rm -rf $OUT/p[0-9]
python3 - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/summary_all.json"))
keep = {k: v for k, v in d.items() if "roundeval" in k}
json.dump(keep, open(sys.argv[1] + "/summary.json", "w"), indent=1, sort_keys=True)
for k, v in keep.items():
    c = {n: x["avg_per_launch"] for n, x in v.items()}
    mf = c.get("SQ_INSTS_MFMA", 0)
    if mf:
        tiles = mf / 96
        cu_cycles = c["GRBM_GUI_ACTIVE"] / 8
        print(k[:50], "launches", v["SQ_INSTS_MFMA"]["launches"], "VALU/tile/SIMD", round(c["SQ_INSTS_VALU"] / tiles / 4, 1), "LDS/tile/CU", round(c["SQ_INSTS_LDS"] / tiles, 1),
              "cycles per tile and CU", round(cu_cycles / (tiles / 256), 1), "LDS busy", round(c["SQ_LDS_IDX_ACTIVE"] / 256 / cu_cycles, 3),
              "MFMA busy", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cu_cycles, 3), "VALU active (4-cycle units)", round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cu_cycles, 3))
PY
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/run_ip_power.py $LOG > $OUT/trace.log 2>&1)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
grep -i "roundeval" $OUT/kernel_stats.csv | cut -c1-200
