#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/deal; rm -rf $O; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_wide.py tests/test_gpu_group_fuzz.py tests/test_gpu_zerocheck.py tests/test_gpu_hal_wide.py tests/test_gpu_hal_coef.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
BNH_PROF=1 BN_GROUP_PROF=1 timeout 300 python tools/bench_keccak_replay.py --log-perms 16 --steps 3 > $O/replay.json 2> $O/prof.txt
grep "eq-set prof\|eq-ind\|group prof\|piop prove" $O/prof.txt | tail -5
timeout 300 python tools/bench_keccak_replay.py --log-perms 16 > $O/keccak_replay.json 2>> $O/err.log; python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/deal/keccak_replay.json"))
print({k:(v.get("ms"),v.get("kernel_ms")) for k,v in d["phases"].items()}, d["total_ms"])
PY
timeout 300 python tools/bench_piop.py claims --n-vars 22 --k 50 > $O/claims_k50.jsonl 2>> $O/err.log; cut -c1-400 $O/claims_k50.jsonl
