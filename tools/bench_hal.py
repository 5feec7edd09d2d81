#!/usr/bin/env python3
"""Timings of the old-HAL entry points (bn_hal_round_evals / bn_hal_fold_multilinear) at prover sizes:
the routed shapes (bivariate product, zerocheck-style a*b + c with an equality indicator), the general interpreter
kernel (a degree-3 composition with interpolation-domain points), the lerp fold in both orders and the switchover
partial evaluation of a B32-packed Transparent multilinear.  One JSON line per case; algorithmic bytes = the stored
evaluations read (+ written)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, default=24)
ap.add_argument("--n-vars-general", type=int, default=20)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
n = 1 << a.n_vars
hal = binius_amd.Context(0, 5 * n + (1 << 16))  # (the general path's rows and temporaries live in context scratch)
alloc = hal.dev_alloc()
d = []
for j in range(3):
    s = alloc.alloc(n)
    step = 1 << 24
    for off in range(0, n, step):
        hal.copy_h2d(synthetic.random_b128_shard(0xA1A0 + j, min(step, n), 1, 0, start=off), s.slice(off, off + min(step, n)))
    d.append(s)
eq = alloc.alloc(n // 2)
hal.copy_d2d(d[2].slice(0, n // 2), eq)
out = alloc.alloc(n)
AB = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])
AB_C = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3)])
ABC_A = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3), ("add", 4, 0)])
ABC = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)])


def timed(name, fn, alg_bytes):
    ts = []
    for _ in range(a.reps + 1):
        hal.sync(); hal.timer_begin(); fn(); ts.append(hal.timer_end_ms())
    ms = min(ts[1:])
    print(json.dumps({"op": name, "ms": round(ms, 4), "alg_GBps": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / 8000, 4)}), flush=True)


full = lambda k, nv: [("folded", d[j].slice(0, 1 << nv), 0) for j in range(k)]
nv = a.n_vars
timed("hal_round_evals a*b at X = 1, inf, High-to-Low, n_vars=%d (routed: matrix-core kernel)" % nv,
      lambda: hal.hal_round_evals(1, nv, None, full(2, nv), [{"composition": AB, "composition_at_infinity": AB, "start": 1, "end": 3, "eq_ind": None}], []), 32 * n)
timed("hal_round_evals (a*b + c) * eq at X = 1, inf, n_vars=%d (the constraint set's two launches: indicator scaling + sums)" % nv,
      lambda: hal.hal_round_evals(1, nv, None, full(3, nv), [{"composition": AB_C, "composition_at_infinity": AB, "start": 1, "end": 3, "eq_ind": eq}], []), 48 * n + 8 * n)
# (c enters at X = 1 only: its upper half -- 40 bytes per point of the cube, not 48)
timed("hal_round_evals a*b + c at X = 1, inf, n_vars=%d (routed; c: a streaming sum of its upper half)" % nv,
      lambda: hal.hal_round_evals(1, nv, None, full(3, nv), [{"composition": AB_C, "composition_at_infinity": AB, "start": 1, "end": 3, "eq_ind": None}], []), 40 * n)
ng = a.n_vars_general
pts = synthetic.random_scalars(5, 1)
timed("hal_round_evals a*b*c + a at X = 1, inf, z (coefficient form: products + one sums launch; BN_HAL_COEF=0: rows + compiled circuits), n_vars=%d" % ng,
      lambda: hal.hal_round_evals(1, ng, None, full(3, ng), [{"composition": ABC_A, "composition_at_infinity": ABC, "start": 1, "end": 4, "eq_ind": None}], pts), 48 << ng)
timed("hal_round_evals a*b*c + a at X = 1, inf, z (coefficient form; BN_HAL_COEF=0: rows + compiled circuits), n_vars=%d" % nv,
      lambda: hal.hal_round_evals(1, nv, None, full(3, nv), [{"composition": ABC_A, "composition_at_infinity": ABC, "start": 1, "end": 4, "eq_ind": None}], pts), 48 << nv)
timed("hal_round_evals a*b at X = 1, inf, High-to-Low, n_vars=%d (routed)" % ng,
      lambda: hal.hal_round_evals(1, ng, None, full(2, ng), [{"composition": AB, "composition_at_infinity": AB, "start": 1, "end": 3, "eq_ind": None}], []), 32 << ng)
timed("hal_round_evals a*b, Low-to-High (round 3: interpreter kernel; now the strided matrix-core kernel), n_vars=%d" % ng,
      lambda: hal.hal_round_evals(0, ng, None, full(2, ng), [{"composition": AB, "composition_at_infinity": AB, "start": 1, "end": 3, "eq_ind": None}], []), 32 << ng)
z = synthetic.random_scalars(6, 1)[0]
timed("hal_fold_multilinear High-to-Low out of place, 2^%d" % nv, lambda: hal.hal_fold_multilinear(1, nv, ("folded", d[0], 0), z, None, out), 24 * n)
timed("hal_fold_multilinear Low-to-High, 2^%d" % nv, lambda: hal.hal_fold_multilinear(0, nv, ("folded", d[0], 0), z, None, out), 24 * n)
timed("hal_fold_multilinear Low-to-High, 2^%d stored of 2^%d (suffix)" % (nv - 2, nv), lambda: hal.hal_fold_multilinear(0, nv, ("folded", d[0].slice(0, n // 4), 7), z, None, out), 24 * (n // 4))
q = alloc.alloc(4)
hal.copy_h2d(synthetic.random_b128(9, 4), q)
packed = d[1].slice(0, n // 4)  # 2^nv B32 values
timed("hal_fold_multilinear switchover: B32-packed Transparent, 2 query variables, 2^%d values -> 2^%d" % (nv, nv - 2),
      lambda: hal.hal_fold_multilinear(1, nv - 1, ("transparent", packed, 5, nv), 0, q, out), 4 * n + 16 * (n // 4))
hal.close()
