#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
bash tools/trace_cmd.sh r5g_claims26 python tools/bench_piop.py claims --n-vars 26 --k 4 --group 1 --steps 2 --warmup 1
bash tools/trace_cmd.sh r5g_bip24 python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --group 1 --steps 2 --warmup 1
bash tools/trace_cmd.sh r5g_piop20 python tools/bench_piop.py piop --n 20 --group 1 --steps 2 --warmup 1
for d in r5g_claims26 r5g_bip24 r5g_piop20; do echo "== $d"; head -12 gpurun_out/$d/kernel_stats.csv | cut -c1-200; done
python3 - <<'PY'
import json
for d in ("r5g_claims26", "r5g_bip24"):
    rows = [json.loads(l) for l in open("gpurun_out/%s/per_launch.jsonl" % d)]
    g = [r for r in rows if "group" in r["kernel"] or "extrapolate" in r["kernel"]]
    print(d, [(r["kernel"][:18], r["us"]) for r in g[-40:]])
PY
