#!/usr/bin/env python3
"""Round-0 evaluation alone (accumulate_kernels of a*b over two n-variable multilinears): time and fraction of the HBM
roofline (algorithmic bytes = 32 * 2^n_vars read once).  BN_GRAM_WAVES=4 selects the four-wave kernel."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd.sumcheck import bivariate_product_expr, round_eval_kernel

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, nargs="+", default=[24, 26])
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
for n_vars in a.n_vars:
    n = 1 << n_vars
    hal = binius_amd.Context(0, 2 * n + 4096)
    alloc = hal.dev_alloc()
    d = []
    for j in range(2):
        s = alloc.alloc(n)
        step = 1 << 24
        for off in range(0, n, step):
            hal.copy_h2d(synthetic.random_b128_shard(0xB1A50000 + j, min(step, n), 1, 0, start=off), s.slice(off, off + min(step, n)))
        d.append(s)
    expr = bivariate_product_expr(hal, 0, 1)
    kernel, maps = round_eval_kernel(n_vars, [1], d, [expr])
    ops, rets, lc = hal.record(kernel, maps)
    ts, val = [], None
    for _ in range(a.reps):
        hal.sync(); hal.timer_begin()
        v = hal.kernel_launch(maps, ops, rets, lc)
        ts.append(hal.timer_end_ms())
        assert val is None or v == val
        val = v
    ms = min(ts[1:])
    print(json.dumps({"op": "round-0 evaluation, n_vars=%d" % n_vars, "waves": os.environ.get("BN_GRAM_WAVES", "8"), "ms": round(ms, 4),
                      "GBps": round(32 * n / ms / 1e6, 1), "frac_of_8TBps": round(32 * n / ms / 1e6 / 8000, 4), "value": [hex(x) for x in val]}))
    hal.close()
