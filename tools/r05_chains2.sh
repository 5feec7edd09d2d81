#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/chains2; rm -rf $O; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_fuzz.py tests/test_gpu_sharded_vs_oracle.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
{ python tools/bench_piop.py claims --n-vars 24 --k 4 --kind piop --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5 --group 1
  python tools/bench_piop.py claims --n-vars 26 --k 4 --kind bipartite --steps 3 --group 1
  BN_GROUP_CHAIN_MIN_LOG2=63 python tools/bench_piop.py claims --n-vars 26 --k 4 --kind bipartite --steps 3 --group 1
  python tools/bench_piop.py claims --n-vars 26 --k 4 --kind piop --steps 3 --group 1
  BN_GROUP_CHAIN_MIN_LOG2=63 python tools/bench_piop.py claims --n-vars 26 --k 4 --kind piop --steps 3 --group 1
  python tools/bench_piop.py piop --n 20 --steps 5 --group 1; } > $O/b.jsonl 2> $O/b.stderr
python - <<'PY'
import json
for l in open('gpurun_out/chains2/b.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    c=d.get('group_counters_one_prove',{})
    print(d.get('bench'), d.get('n_vars') or d.get('n_varss'), d.get('kind'), d.get('ms_per_prove') or d.get('prove_ms'), {k:c.get(k) for k in ('launches','jobs_fused','jobs_eval','jobs_fold','chains','prefolds')}, d.get('prof_ms'))
PY
