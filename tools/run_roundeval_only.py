#!/usr/bin/env python3
"""Runs only the r = n round-eval (and optionally fold) kernels a few times -- a target for rocprofv3."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd.sumcheck import bivariate_product_expr, round_eval_kernel

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, default=24)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--fold", action="store_true")
a = ap.parse_args()
n = 1 << a.n_vars
hal = binius_amd.Context(0, 3 * n + 4096)
alloc = hal.dev_alloc()
d = []
for j in range(2):
    s = alloc.alloc(n); hal.copy_h2d(synthetic.random_b128(0xB1A50000 + j, n), s); d.append(s)
f = alloc.alloc(n // 2)
expr = bivariate_product_expr(hal, 0, 1)
kernel, maps = round_eval_kernel(a.n_vars, [1], d, [expr])
ops, rets, lc = hal.record(kernel, maps)
for _ in range(a.reps):
    hal.kernel_launch(maps, ops, rets, lc)
    if a.fold:
        e0, e1 = d[0].split_half()
        hal.copy_d2d(e0, f)
        hal.extrapolate_line(f, e1, 12345678901234567890123)
hal.sync()
hal.close()
