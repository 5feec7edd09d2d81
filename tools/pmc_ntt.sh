#!/bin/bash
# VALU / LDS occupancy of the NTT kernels from counters-only rocprofv3 passes over tools/profile_ntt.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ntt
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $R/tools/profile_ntt.py --reps 2 > $OUT/p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 > /dev/null
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "ntt" not in k: continue
    w = v["SQ_WAVES"]["avg_per_launch"]
    cyc = v["SQ_WAVE_CYCLES"]["avg_per_launch"] * 4 / w
    print(k[:70], "launches", v["SQ_WAVES"]["launches"], "waves", w, "wave-resident cycles", round(cyc),
          "VALU instr/wave", round(v["SQ_INSTS_VALU"]["avg_per_launch"] / w), "VALU issue busy (4 cyc/instr, waves per SIMD = %.1f)" % (w / 1024),
          round(v["SQ_INSTS_VALU"]["avg_per_launch"] / 1024 * 4 / (cyc * max(1.0, w / 1024 / (w / 1024))), 3),
          "LDS busy", round(v["SQ_LDS_IDX_ACTIVE"]["avg_per_launch"] / 256 / cyc, 3), "bank conflict cycles", v["SQ_LDS_BANK_CONFLICT"]["avg_per_launch"])
PY
