#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5k
export TMPDIR=/tmp
BN_GROUP_ADJ=1 timeout 900 python -m pytest tests/test_gpu_group.py tests/test_gpu_group_fuzz.py -q -m gpu -x -k "not 20- and not 22- and not 17" > gpurun_out/r5k/tests_adj.log 2>&1
tail -3 gpurun_out/r5k/tests_adj.log
timeout 900 python -m pytest tests/test_gpu_hal.py tests/test_gpu_circuits.py -q -m gpu -x > gpurun_out/r5k/tests_hal.log 2>&1
tail -3 gpurun_out/r5k/tests_hal.log
{
for rep in 1 2 3; do for A in 0 1; do
echo "ADJ=$A"
BN_GROUP_ADJ=$A python tools/bench_piop.py claims --n-vars 26 --k 4 --steps 3 --group 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_prove'], d['prof_ms'])"
BN_GROUP_ADJ=$A python tools/bench_piop.py claims --n-vars 24 --k 4 --steps 5 --group 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_prove'], d['prof_ms'])"
BN_GROUP_ADJ=$A python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5 --group 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_prove'], d['prof_ms'])"
done; done
python tools/bench_hal.py | grep "a\*b\*c"
} > gpurun_out/r5k/adj.log 2>&1
cat gpurun_out/r5k/adj.log
