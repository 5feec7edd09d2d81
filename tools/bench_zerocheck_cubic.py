#!/usr/bin/env python3
"""EqIndSumcheckProver (binius_amd/host/eq_ind.hpp) over a small constraint set with constraints of degree 3: a*b*c + d*e + f,
a*b + c, a + b + e, a*b*c over six columns of 2^n elements -- the round evaluations at X = 1, infinity and the domain point 2 are the
old HAL's coefficient-form requests (BN_HAL_COEF=0: the general code).  One JSON line; the verifier's equations on
what the device produced."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd._host import EqIndPlan

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, default=20)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
F = binius_amd.HostField
n = 1 << a.n_vars
abc = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)]
comps = [(abc + [("var", 3), ("var", 4), ("mul", 5, 6), ("add", 4, 7), ("var", 5), ("add", 8, 9)], abc),
         ([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3)], [("var", 0), ("var", 1), ("mul", 0, 1)]),
         ([("var", 0), ("var", 1), ("add", 0, 1), ("var", 4), ("add", 2, 3)],) * 2, (abc, abc)]
degrees = [3, 2, 1, 3]
stream = synthetic.random_scalars(0xC0B1C, 2 * a.n_vars + 1 + len(comps))
eqc, ch, bc, sums = stream[: a.n_vars], stream[a.n_vars : 2 * a.n_vars], stream[2 * a.n_vars], stream[2 * a.n_vars + 1 :]
best = None
with binius_amd.Context(0, 8 * n + (1 << 16)) as hal:
    for step in range(a.steps + 1):
        alloc = hal.dev_alloc()
        d = []
        for j in range(6):
            s = alloc.alloc(n)
            for off in range(0, n, 1 << 22):
                m = min(1 << 22, n - off)
                hal.copy_h2d(synthetic.random_b128_shard(0x51 + j, m, 1, 0, start=off), s.slice(off, off + m))
            d.append(s)
        scratch = alloc.alloc(n // 2 + 64)
        plan = EqIndPlan(hal, a.n_vars, d, comps, sums, eqc, scratch, bc, ch, degrees)
        hal.sync()
        t0 = time.perf_counter()
        plan.run()
        hal.sync()
        dt = (time.perf_counter() - t0) * 1e3
        if step and (best is None or dt < best):
            best = dt
    coeffs = plan.round_coeffs()
# the verifier's side: P(0) + P(1) = the running claim, round by round
running = 0
p = 1
for s_ in sums:
    running ^= F.mul(p, s_)
    p = F.mul(p, bc)
ok = True
for r in range(a.n_vars):
    c = coeffs[r]
    p1 = 0
    for v in c:
        p1 ^= v
    ok = ok and (c[0] ^ p1) == running
    acc = 0
    for v in reversed(c):
        acc = F.mul(acc, ch[r]) ^ v
    running = acc
print(json.dumps({"bench": "zerocheck, constraints of degree 3 / 2 / 1 / 3 over 6 columns", "n_vars": a.n_vars, "coef_path": os.environ.get("BN_HAL_COEF", "1") != "0",
                  "ms_per_prove": round(best, 3), "round_sums_check": ok}))
