#!/usr/bin/env python3
"""A few fused fold + evaluation launches at one size (2^LOG elements per array, default 27), nothing else of note on the device --
for counter passes (tools/pmc_fused.sh): extrapolate_line_batch is deferred by the ABI and runs fused with the round evaluation
that reads the folded arrays (k_foldeval_mfma_fp4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << lg
hal = binius_amd.Context(0, 2 * n + 4096)
alloc = hal.dev_alloc()
bufs = [alloc.alloc(n) for _ in range(2)]
step = 1 << 22
for j, b in enumerate(bufs):
    for off in range(0, n, step):
        hal.copy_h2d(synthetic.random_b128_shard(0xFE00 + j, step, 1, 0, start=off), b.slice(off, off + step))
expr = bivariate_product_expr(hal, 0, 1)
zs = synthetic.random_scalars(0xFE, reps)
half = n // 2
for r in range(reps):
    # (the arrays are folded in place and then re-used at full length: the values do not matter, the launch shape does)
    hal.extrapolate_line_batch([b.slice(0, half) for b in bufs], [b.slice(half, n) for b in bufs], zs[r])
    calculate_round_evals(hal, lg - 1, [1], [b.slice(0, half) for b in bufs], [expr])
hal.sync()
hal.close()
