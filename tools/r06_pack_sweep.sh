#!/bin/bash
# NEEDS a measurement build (make -C binius_amd/csrc clean && make -C binius_amd/csrc BN_KNOBS=1): the shipped library ignores the knob and says so.
# round 6: the super-unit count of packed group launches (kernels_group.hip launch_group), forced through BN_GROUP_PACK_U:
# fused-launch time of a 50-claim prover at 2^22 by U (-1: one unit per job)
for u in -1 1 2 4 8 16 32; do
  echo "== U $u"
  BN_GROUP_PACK_U=$u python tools/bench_piop.py claims --k ${K:-50} --kind ${KIND:-disjoint} --n-vars ${N:-22} --group 1 --steps 2 --warmup 1 2>&1 | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print({k:r[k] for k in ('ms_per_prove','prof_ms','group_launch_frac')})"
done
