#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5h
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_fuzz.py -q -m gpu -x -k "not 20- and not 22- and not 17" > gpurun_out/r5h/tests.log 2>&1
tail -3 gpurun_out/r5h/tests.log
bash tools/trace_cmd.sh r5h_claims26 python tools/bench_piop.py claims --n-vars 26 --k 4 --group 1 --steps 2 --warmup 1
bash tools/trace_cmd.sh r5h_bip24 python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --group 1 --steps 2 --warmup 1
python3 - <<'PY'
import json
for d in ("r5h_claims26", "r5h_bip24"):
    rows = [json.loads(l) for l in open("gpurun_out/%s/per_launch.jsonl" % d)]
    g = [r for r in rows if "group_fp4" in r["kernel"] or "extrapolate" in r["kernel"]]
    print(d, [(r["kernel"][4:14], r["us"]) for r in g[-30:]])
PY
{
python tools/bench_piop.py claims --n-vars 24 --k 4 --steps 5 --group 1
python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5 --group 1
python tools/bench_piop.py claims --n-vars 26 --k 4 --steps 3 --group 1
python tools/bench_piop.py piop --n 20 --steps 5 --group 1
} > gpurun_out/r5h/bench.log 2>&1
grep -v "^\[" gpurun_out/r5h/bench.log | cut -c1-420
