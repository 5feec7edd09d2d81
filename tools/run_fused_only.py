#!/usr/bin/env python3
"""Runs the r = n round evaluation and then the fused fold(r) + evaluation(r-1) kernel a few times
-- a target for rocprofv3 (kernel trace or one --pmc set per run).  The fold is the prover's first
one (copy evals_0 into a fresh buffer, fold that), so the inputs are never modified."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, default=24)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n = 1 << a.n_vars
hal = binius_amd.Context(0, 3 * n + 4096)
alloc = hal.dev_alloc()
d = []
for j in range(2):
    s = alloc.alloc(n); hal.copy_h2d(synthetic.random_b128(0xB1A50000 + j, n), s); d.append(s)
dst = [alloc.alloc(n // 2) for _ in range(2)]
expr = bivariate_product_expr(hal, 0, 1)
halves = [x.split_half() for x in d]
for _ in range(a.reps):
    calculate_round_evals(hal, a.n_vars, [1], d, [expr])
    for (lo, _), t in zip(halves, dst):
        hal.copy_d2d(lo, t)
    hal.extrapolate_line_batch(dst, [hi for _, hi in halves], 12345678901234567890123)
    calculate_round_evals(hal, a.n_vars - 1, [1], dst, [expr])
hal.sync()
hal.close()
