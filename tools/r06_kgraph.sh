#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kgraph; rm -rf $O; mkdir -p $O; cd $R
BN_GROUP_PROF=1 BNH_PROF=1 timeout 300 python tools/bench_piop.py claims --n-vars 22 --k 100 --kind keccak --steps 5 --group 1 > $O/claims.jsonl 2> $O/prof.txt
cut -c1-1200 $O/claims.jsonl; tail -4 $O/prof.txt | cut -c1-700
bash tools/trace_cmd.sh kgraph/trace python tools/bench_piop.py claims --n-vars 22 --k 100 --kind keccak --steps 1 --warmup 1 --group 1
tail -40 $O/trace/per_launch.jsonl
