#!/bin/bash
# rebuilds the stand-alone measurement binaries under tools/ from the CURRENT kernel sources (they include csrc/*.hip
# directly, so a binary built before a kernel change measures the old kernel); hipcc cross-compiles without a GPU
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude"
set -e
for t in launch_latency coop_latency signal_latency small_round_phases two_round_phases mfma_round_phases gram_bench mfma_issue valu_rate bsmul_rate fold_variants fp4_probe mfma_gram; do
  [ -f tools/$t.hip ] && $H tools/$t.hip -o tools/$t &
done
for v in 0 1 3 7; do $H -DFE_VARIANT=$v tools/gram_bench.hip -o tools/gram_bench_v$v & done
$H -DBN_FP4_PHASES=1 tools/gram_bench.hip -o tools/gram_bench_ph &
wait
ls -la tools | grep -E "^-rwx" | awk '{print $9, $6, $7, $8}'
