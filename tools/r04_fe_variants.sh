#!/bin/bash
# the fused fold + evaluation kernel in the FE_VARIANT forms of ctable_mul_acc (tools/gram_bench.hip built with -DFE_VARIANT=v):
# bit-exactness on the small sizes, then isolated and sustained launch times at 2^26 and 2^27 elements per array
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fe_variants
mkdir -p $O
cd $R
for rep in 1 2; do
for v in 0 1 3 7; do
  echo "== FE_VARIANT=$v (rep $rep)"
  tools/gram_bench_v$v 18 2>&1 | grep -E "fused N=|ALL OK|FAILED" | head -8
  for l in 26 27; do tools/gram_bench_v$v $l prof 40 2>&1 | grep -E "fused N=" | tail -6; done
done
done > $O/times.txt 2>&1
cat $O/times.txt
