#!/bin/bash
# A/B of round 4's second batch of kernel changes on one box: the NTT's register-resident passes (BN_NTT_REG_PASS) and the
# element-wise product's two batches per rebuild (BN_MUL9_DUAL)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/ab2
mkdir -p $O
cd $R
{
for rep in 1 2; do
  for v in 1 0; do
    echo "== BN_NTT_REG_PASS=$v (rep $rep)"
    BN_NTT_REG_PASS=$v python tools/profile_ntt.py --reps 3 2>&1 | grep -E "forward NTT" | tail -2
    BN_NTT_REG_PASS=$v python tools/profile_ntt.py --reps 3 --log-n 22 --elem-level 7 2>&1 | grep -E "forward NTT" | tail -1
  done
  for v in 1 0; do
    echo "== BN_MUL9_DUAL=$v (rep $rep)"
    BN_MUL9_DUAL=$v python tools/bench_ops.py 2>&1 | grep -E "compute_composite|a\*b\*eq|generic|pairwise"
    BN_MUL9_DUAL=$v python tools/bench_hal.py 2>&1 | grep -E "eq at|a\*b\*c"
  done
done
} > $O/times.txt 2>&1
cat $O/times.txt
