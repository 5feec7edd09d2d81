#!/usr/bin/env python3
"""End-to-end time of the MLE-check prover mirror (compiled host loop, bnh_bivariate_mlecheck_prove):
n-variable eq-indicator sumcheck of one bivariate product over m = 2 multilinears."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd._host import MlecheckPlan
from binius_amd.sumcheck import eq_ind_partial_eval

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, default=24)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
n_vars, m = a.n_vars, 2
n = 1 << n_vars
if __import__("os").environ.get("BN_BIND_NUMA") != "0":
    binius_amd.bind_host_thread_to_device(0)  # (the driving thread on the device's NUMA node)
hal = binius_amd.Context(0, m * n + (m + 2) * (n // 2) + 4096)
alloc = hal.dev_alloc()
d = []
for j in range(m):
    s = alloc.alloc(n); hal.copy_h2d(synthetic.random_b128(0xB1A50000 + j, n), s); d.append(s)
eq_ch = synthetic.random_scalars(0xE9, n_vars)
eq = eq_ind_partial_eval(hal, alloc, eq_ch[: n_vars - 1])
scratch = alloc.alloc((m + 1) * (n // 2))
stream = synthetic.random_scalars(0xC4A2, n_vars + 1)
plan = MlecheckPlan(hal, n_vars, d, eq, eq_ch, scratch, [(0, 1)], [0], stream[0], stream[1:])
for _ in range(2):
    plan.run()
hal.sync()
t0 = time.perf_counter()
for _ in range(a.steps):
    plan.run()
hal.sync()
ms = (time.perf_counter() - t0) * 1e3 / a.steps
hal.prof_begin()
plan.run()
prof = {k: {"ms": round(v[0], 4), "launches": v[1]} for k, v in hal.prof_end().items() if v[1]}
print(json.dumps({"op": "bivariate MLE-check prove, n_vars=%d, m=2" % n_vars, "prover": "weighted" if plan.last_mode() == 1 else "literal",
                  "ms": round(ms, 4), "elems_per_s": round(m * n / ms * 1e3), "kernels": prof}))
