#!/usr/bin/env python3
"""Throughput of several INDEPENDENT 2^n-variable sumchecks sharing one GPU: one host thread, context and
stream per prover (the compiled prover loop releases the GIL).  The latency-bound small rounds and launch
round trips of one prover overlap with the bandwidth-bound rounds of the others -- how a prover with many
claims keeps the device busy.  Not the bench.py metric (that one is a single sumcheck at a time)."""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd._host import SumcheckPlan
from binius_amd._ffi import HostField as F

ap = argparse.ArgumentParser()
ap.add_argument("--n-vars", type=int, default=24)
ap.add_argument("--provers", type=int, nargs="*", default=[1, 2, 3, 4])
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
n, m = 1 << a.n_vars, 2
stream = synthetic.random_scalars(0xC4A1, a.n_vars + 1)
batch_coeff, challenges = stream[0], stream[1:]
host = [synthetic.random_b128(0xB1A50000 + j, n) for j in range(m)]


def make(idx):
    hal = binius_amd.Context(0, 3 * n + (1 << 12))
    alloc = hal.dev_alloc()
    d = []
    for x in host:
        s = alloc.alloc(n)
        hal.copy_h2d(x, s)
        d.append(s)
    claim = hal.inner_product(d[0], 7, d[1])
    plan = SumcheckPlan(hal, a.n_vars, d, alloc.alloc(m * n // 2), [(0, 1)], [claim], batch_coeff, challenges)
    plan.run()
    return hal, plan


for P in a.provers:
    ctxs = [make(i) for i in range(P)]
    ref = ctxs[0][1].round_coeffs()
    go = threading.Barrier(P + 1)

    def work(plan):
        go.wait()
        for _ in range(a.steps):
            plan.run()

    ths = [threading.Thread(target=work, args=(pl,)) for _, pl in ctxs]
    for t in ths: t.start()
    for h, _ in ctxs: h.sync()
    go.wait(); t0 = time.perf_counter()
    for t in ths: t.join()
    for h, _ in ctxs: h.sync()
    dt = time.perf_counter() - t0
    ok = all(pl.round_coeffs() == ref for _, pl in ctxs)
    print(json.dumps({"op": "%d concurrent 2^%d-var sumchecks on one GPU" % (P, a.n_vars), "ms_per_sumcheck": round(dt * 1e3 / (P * a.steps), 4),
                      "G_elems_per_s": round(m * n * P * a.steps / dt / 1e9, 2), "identical_transcripts": ok}), flush=True)
    for h, _ in ctxs: h.close()
