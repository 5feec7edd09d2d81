#!/bin/bash
# chains on / off: group tests, then the claim-shape benches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/chains; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_fuzz.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for C in 0 63; do
  export BN_GROUP_CHAIN_MIN_LOG2=$C
  { python tools/bench_piop.py claims --n-vars 24 --k 4 --kind piop --steps 5 --group 1
    python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5 --group 1
    python tools/bench_piop.py claims --n-vars 20 --k 4 --kind bipartite --steps 10 --group 1
    python tools/bench_piop.py claims --n-vars 20 --k 4 --kind piop --steps 10 --group 1
    python tools/bench_piop.py claims --n-vars 24 --k 4 --steps 5 --group 1
    python tools/bench_piop.py claims --n-vars 16 --k 4 --kind bipartite --steps 20 --group 1
    python tools/bench_piop.py piop --n 20 --steps 5 --group 1
    python tools/bench_piop.py piop --n 16 --steps 10 --group 1; } > $O/chains_$C.jsonl 2> $O/chains_$C.stderr
done
python - <<'PY'
import json
for C in (0,63):
    for l in open('gpurun_out/chains/chains_%d.jsonl'%C):
        try: d=json.loads(l)
        except Exception: continue
        c=d.get('group_counters_one_prove',{})
        print(C, d.get('bench'), d.get('n_vars') or d.get('n_varss'), d.get('kind'), d.get('ms_per_prove') or d.get('prove_ms'), {k:c.get(k) for k in ('launches','jobs_fused','jobs_eval','jobs_fold','chains','prefolds')})
PY
