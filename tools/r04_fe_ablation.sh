#!/bin/bash
# the fused fold + evaluation kernel with its arithmetic taken out piece by piece (tools/gram_bench.hip built with -DFE_ABL=x, see
# kernels_foldeval_mfma.hip): sustained launch times at 2^27 (and 2^25) elements per array.  Build first: tools/r04_fe_ablation.sh build
R=${GRAFT_REPO_ROOT:-.}
cd $R
ABLS="0 4 8 12 1 2 3 13 14 19 51 63"
if [ "$1" = build ]; then
  H="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude"
  for a in $ABLS; do $H -DFE_ABL=$a tools/gram_bench.hip -o tools/gram_bench_abl$a 2>/dev/null & done
  wait
  ls tools/gram_bench_abl*
  exit 0
fi
O=$R/gpurun_out/fe_ablation
mkdir -p $O
for rep in 1 2; do
for a in $ABLS; do
  echo "== FE_ABL=$a (rep $rep)"
  for l in 25 27; do timeout 120 tools/gram_bench_abl$a $l prof 30 2>&1 | grep -E "fused N=.*sustained" | tail -2; done
done
done > $O/times.txt 2>&1
cat $O/times.txt
