#!/bin/bash
# k = 50 / m = 100 / n = 22: host phases of the group path (BN_GROUP_PROF, BNH_PROF) and the hosting threshold, one box
mkdir -p gpurun_out/k50
for i in 1 2 3; do
  python tools/bench_piop.py claims --n-vars 22 --k 50 --group 1 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', d['ms_per_prove'], d['whole_prove_frac_of_64mN'])"
done
BN_GROUP_PROF=1 BNH_PROF=1 python tools/bench_piop.py claims --n-vars 22 --k 50 --group 1 --steps 20 --warmup 3 > gpurun_out/k50/prof.out 2> gpurun_out/k50/prof.err
tail -3 gpurun_out/k50/prof.err | cut -c1-900
for w in 12 13 14 15 16 17; do
  BN_GROUP_HT_WORK_LOG2=$w python tools/bench_piop.py claims --n-vars 22 --k 50 --group 1 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ht_work_log2', $w, d['ms_per_prove'], d['group_counters_one_prove']['hosted_evals'])"
done
