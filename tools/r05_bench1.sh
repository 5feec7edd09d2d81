#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5b
export TMPDIR=/tmp
{
python tools/bench_piop.py claims --n-vars 20 --k 4 --steps 10
python tools/bench_piop.py claims --n-vars 24 --k 4 --steps 5
python tools/bench_piop.py claims --n-vars 24 --k 1 --steps 5 --group 1
python tools/bench_piop.py claims --n-vars 24 --k 4 --kind piop --steps 5
python tools/bench_piop.py claims --n-vars 24 --k 4 --kind bipartite --steps 5
python tools/bench_piop.py claims --n-vars 26 --k 4 --steps 3 --group 1
python tools/bench_piop.py claims --n-vars 16 --k 4 --steps 20
python tools/bench_piop.py claims --n-vars 12 --k 8 --steps 20
python tools/bench_piop.py piop --n 20 --steps 5
python tools/bench_piop.py piop --n 12 --steps 10
python bench.py --n-vars 24 --steps 10 --warmup 3 --no-cpu-baseline
python bench.py --steps 10 --warmup 3 --no-cpu-baseline
} > gpurun_out/r5b/bench.log 2>&1
grep -v "^\[" gpurun_out/r5b/bench.log | cut -c1-900
