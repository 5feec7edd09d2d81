#!/usr/bin/env python3
"""The old HAL's degree-3 requests alone: a*b*c + a at X = 1, infinity and --points domain points, with or without an
equality indicator, High-to-Low over full multilinears -- the coefficient-form path (BN_HAL_COEF=0: rows + compiled circuits).
One JSON line per size; algorithmic bytes = the stored evaluations read."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, nargs="+", default=[24, 20])
ap.add_argument("--points", type=int, default=1)
ap.add_argument("--eq", type=int, default=0)
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
nmax = 1 << max(a.n_vars)
hal = binius_amd.Context(0, 8 * nmax + (1 << 16))
alloc = hal.dev_alloc()
d = []
for j in range(3):
    s = alloc.alloc(nmax)
    step = 1 << 24
    for off in range(0, nmax, step):
        hal.copy_h2d(synthetic.random_b128_shard(0xA1A0 + j, min(step, nmax), 1, 0, start=off), s.slice(off, off + min(step, nmax)))
    d.append(s)
eq = alloc.alloc(nmax // 2)
hal.copy_d2d(d[2].slice(0, nmax // 2), eq)
ABC_A = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3), ("add", 4, 0)])
ABC = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)])
pts = synthetic.random_scalars(5, a.points)
for nv in a.n_vars:
    mls = [("folded", d[j].slice(0, 1 << nv), 0) for j in range(3)]
    ev = [{"composition": ABC_A, "composition_at_infinity": ABC, "start": 1, "end": 3 + a.points, "eq_ind": eq.slice(0, 1 << (nv - 1)) if a.eq else None}]
    ts = []
    for _ in range(a.reps + 1):
        hal.sync(); hal.timer_begin(); hal.hal_round_evals(1, nv, None, mls, ev, pts); ts.append(hal.timer_end_ms())
    ms = min(ts[1:])
    alg = (48 + (8 if a.eq else 0)) << nv
    print(json.dumps({"op": "hal_round_evals %sa*b*c + a%s at X = 1, inf + %d domain point(s), n_vars=%d" % ("(" if a.eq else "", ") * eq" if a.eq else "", a.points, nv),
                      "coef_path": os.environ.get("BN_HAL_COEF", "1") != "0", "ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1),
                      "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4), "group_launches": hal.group_counters()["launches"]}), flush=True)
hal.close()
