#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/coef2; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_hal_coef.py tests/test_gpu_zerocheck.py tests/test_gpu_hal_wide.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for eq in 0 1; do for p in 1 3; do
timeout 200 python tools/bench_hal_cubic.py --eq $eq --points $p >> $O/cubic.jsonl 2>> $O/err.log
BN_HAL_COEF=0 timeout 200 python tools/bench_hal_cubic.py --eq $eq --points $p >> $O/cubic.jsonl 2>> $O/err.log
done; done
cat $O/cubic.jsonl
bash tools/trace_cmd.sh coef2/trace python tools/bench_hal_cubic.py --n-vars 24 --reps 2
tail -12 $O/trace/per_launch.jsonl
timeout 300 python tools/bench_keccak_replay.py --log-perms 16 > $O/keccak_replay.json 2>> $O/err.log; python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/coef2/keccak_replay.json"))
print({k:(v.get("ms"),v.get("kernel_ms")) for k,v in d["phases"].items()}, d["total_ms"])
PY
