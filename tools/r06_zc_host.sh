#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/zc_host; rm -rf $O; mkdir -p $O; cd $R
BNH_PROF=1 BN_GROUP_PROF=1 timeout 300 python tools/bench_keccak_replay.py --log-perms 16 --steps 3 > $O/replay.json 2> $O/prof.txt
grep -v "^\s*$" $O/prof.txt | tail -40
