#!/bin/bash
# counters of the old HAL's degree-3 request in coefficient form: k_mul9_jobs_dual, k_group_fp4, k_xor_sum
R=$GRAFT_REPO_ROOT; cd $R
bash tools/pmc_cmd.sh pmc_cubic python tools/bench_hal_cubic.py --n-vars 24 --reps 2 > /dev/null 2>&1
python3 - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_cubic/summary.json"))
keep = {k: v for k, v in d.items() if "mul9_jobs" in k or "k_group_fp4" in k or "xor_sum" in k}
json.dump(keep, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_cubic/hal_cubic_pmc.json", "w"), indent=1, sort_keys=True)
for k, v in keep.items():
    print(k[:50], {c: x["avg_per_launch"] for c, x in v.items() if c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_ACTIVE_INST_LDS")})
PY
