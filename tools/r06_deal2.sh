#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/deal2; rm -rf $O; mkdir -p $O; cd $R
BNH_PROF=1 BN_GROUP_PROF=1 timeout 300 python tools/bench_keccak_replay.py --log-perms 16 --steps 3 > $O/replay.json 2> $O/prof.txt
grep "eq-set prof\|eq-ind\|group prof\|piop prove" $O/prof.txt | tail -5
