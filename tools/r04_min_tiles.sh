#!/bin/bash
# the smallest round the matrix-core kernels take (BN_MFMA_MIN_TILES: default 2 x CUs = 512 tiles = 2^17 points): step times with
# the r = 18 (and r = 17) rounds moved from the 9-lane kernels onto them
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/min_tiles
mkdir -p $O
cd $R
one() { # name, env..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py "$@" --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'ms_per_step', round(d['ms_per_step'],4), d['verifier_check'], d['transcript_digest'])"
}
{
for rep in 1 2; do
  for n in 20 24 25; do
    for mt in 512 256 128; do
      one "n=$n BN_MFMA_MIN_TILES=$mt" BN_MFMA_MIN_TILES=$mt -- --n-vars $n --steps 20 --warmup 3
    done
  done
done
} > $O/step_times.txt 2>&1
cat $O/step_times.txt
