#!/usr/bin/env python3
"""Per-size kernel timings (hipEvent, on the context's stream): fold and round-eval at every
round size r = n..1, reporting algorithmic GB/s and fraction of the 8 TB/s HBM roofline."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-vars", type=int, default=24)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--min-r", type=int, default=1)
    args = ap.parse_args()
    import binius_amd
    from binius_amd import synthetic
    from binius_amd.sumcheck import bivariate_product_expr, round_eval_kernel

    n = 1 << args.n_vars
    hal = binius_amd.Context(0, 2 * n + 2 * (n // 2) + 4096)
    alloc = hal.dev_alloc()
    d = []
    for j in range(2):
        s = alloc.alloc(n)
        hal.copy_h2d(synthetic.random_b128(0xB1A50000 + j, n), s)
        d.append(s)
    scratch = [alloc.alloc(n // 2) for _ in range(2)]
    expr = bivariate_product_expr(hal, 0, 1)
    z = synthetic.random_scalars(0xC4A1, 1)[0]
    rows = []
    for r in range(args.n_vars, args.min_r - 1, -1):
        N = 1 << r
        mls = [x.slice(0, N) for x in d]
        kernel, maps = round_eval_kernel(r, [1], mls, [expr])
        ops, rets, lc = hal.record(kernel, maps)
        hal.kernel_launch(maps, ops, rets, lc)
        hal.prof_begin()
        for _ in range(args.reps):
            hal.kernel_launch(maps, ops, rets, lc, want_host=True)
        p = hal.prof_end()
        re_tot = p["round_eval"][0] + p["round_eval_mfma"][0]
        re_ms = re_tot / (p["round_eval"][1] + p["round_eval_mfma"][1])
        # fold of one multilinear of size N into scratch (copy first so inputs stay intact)
        e0, e1 = mls[0].split_half()
        f = scratch[0].slice(0, N // 2)
        hal.copy_d2d(e0, f)
        hal.extrapolate_line(f, e1, z)
        hal.prof_begin()
        for _ in range(args.reps):
            hal.extrapolate_line(f, e1, z)
        p = hal.prof_end()
        fo_ms = p["fold"][0] / p["fold"][1]
        re_gbs = 16 * 2 * N / (re_ms * 1e-3) / 1e9
        fo_gbs = 24 * N / (fo_ms * 1e-3) / 1e9
        rows.append({"r": r, "round_eval_ms": round(re_ms, 5), "round_eval_GBps": round(re_gbs, 1), "round_eval_frac": round(re_gbs / 8000, 4),
                     "fold_ms": round(fo_ms, 5), "fold_GBps": round(fo_gbs, 1), "fold_frac": round(fo_gbs / 8000, 4)})
        print(json.dumps(rows[-1]), flush=True)
    hal.close()


if __name__ == "__main__":
    main()
