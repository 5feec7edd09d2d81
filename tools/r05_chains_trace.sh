#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for C in 0 63; do
  BN_GROUP_CHAIN_MIN_LOG2=$C tools/trace_cmd.sh ct_bip_$C python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --group 1 --steps 1 --warmup 1 > /dev/null 2>&1
  cd $R
  python - <<PY
import json
rows=[json.loads(l) for l in open('gpurun_out/ct_bip_$C/per_launch.jsonl')]
rows=[r for r in rows if 'group_fp4' in r['kernel'] or 'extrapolate' in r['kernel']]
print($C, [(r['kernel'][4:16], r['us'], r['grid']) for r in rows[-24:]])
PY
done
