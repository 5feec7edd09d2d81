#!/usr/bin/env python
"""Times the call shape of the reference's PCS prover: (a) ONE BivariateSumcheckProver with k product claims
over m multilinears, (b) the front-loaded batch / piop::prove with FRI interleaved -- with the claim-group path on and off
(BN_GROUP), per-class kernel time from the context's hipEvents (bn_prof_*), the fused launches against the HBM roofline
(algorithmic bytes 24 * m * 2^r per fused round), and the verifier's equations on what was timed.

  python tools/bench_piop.py claims --n-vars 24 --k 4 [--kind disjoint|piop|bipartite] [--steps 5]
  python tools/bench_piop.py piop --n 20 [--log-inv-rate 1 --log-batch 4 --arity 4]

Inputs are generated ON the device (tensor expansions of random points: dense, distinct field elements -- the kernels' clock
depends on the data) so that large instances need no host memory; parity against the oracle at these shapes is
tests/test_gpu_group.py's job (n <= 22), here only the verifier's equations are checked.  One JSON line per configuration."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK = 8.0e12


def evaluate_univariate(F, coeffs, x):
    e = 0
    for c in reversed(coeffs):
        e = F.mul(e, x) ^ c
    return e


def device_random(hal, alloc, F, seed, n_vars):
    """2^n_vars dense pseudo-random elements without a host copy: the tensor expansion of a random point."""
    from binius_amd import synthetic

    out = alloc.alloc(1 << n_vars)
    hal.fill(out.slice(0, 1), 1 + seed)
    hal.tensor_expand(0, synthetic.random_scalars(seed, n_vars), out)
    return out


def claims_for(kind, k):
    if kind == "disjoint":
        return 2 * k, [(i, k + i) for i in range(k)]
    if kind == "piop":
        return 2 * k, [(i, k + i) for i in range(k - 1)] + [(0, 2 * k - 1)]
    if kind == "star":  # k committed multilinears against ONE transparent
        return k + 1, [(i, k) for i in range(k)]
    if kind == "keccak":  # k committed columns against three transparents: every column at one point, half at a second, a quarter at a third
        c, t = k, 3
        return c + t, [(i, c + i % t) for i in range(c)] + [(i, c + (i + 1) % t) for i in range(0, c, 2)] + [(i, c + (i + 2) % t) for i in range(0, c, 4)]
    c = int(round(k ** 0.5))
    return 2 * c, [(i, c + j) for i in range(c) for j in range(c)]


def run_claims(args):
    import binius_amd
    from binius_amd import synthetic
    from binius_amd._host import SumcheckPlan

    F = binius_amd.HostField
    n_vars, k = args.n_vars, args.k
    m, comps = claims_for(args.kind, k)
    n = 1 << n_vars
    out = []
    for group in ([1, 0] if args.group < 0 else [args.group]):
        os.environ["BN_GROUP"] = str(group)
        pad = int(os.environ.get("BENCH_PAD_ELEMS", "0"))  # (experiment: arrays NOT at power-of-two strides from each other)
        with binius_amd.Context(0, m * n + m * (n // 2) + 4096 + m * pad) as hal:
            alloc = hal.dev_alloc()
            d = []
            for j in range(m):
                d.append(device_random(hal, alloc, F, 0xB1A5 + j, n_vars))
                if pad:
                    alloc.alloc(pad)
            scratch = alloc.alloc(m * (n // 2))
            sums = [hal.inner_product(d[i], 7, d[j]) for i, j in comps]
            stream = synthetic.random_scalars(0xC4A1, n_vars + 1)
            bc, ch = stream[0], stream[1:]
            plan = SumcheckPlan(hal, n_vars, d, scratch, comps, sums, bc, ch)
            for _ in range(args.warmup):
                plan.run()
            hal.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                plan.run()
            hal.sync()
            ms = (time.perf_counter() - t0) * 1e3 / args.steps
            # verifier's equations on the timed transcript
            coeffs, finals = plan.round_coeffs(), plan.final_evals()
            running = evaluate_univariate(F, sums, bc)
            ok = True
            for r in range(n_vars):
                c = coeffs[r]
                ok = ok and (c[0] ^ (c[0] ^ c[1] ^ c[2])) == running
                running = evaluate_univariate(F, c, ch[r])
            acc, p = 0, 1
            for i, j in comps:
                acc ^= F.mul(p, F.mul(finals[i], finals[j]))
                p = F.mul(p, bc)
            ok = ok and acc == running
            c0 = hal.group_counters()
            hal.prof_begin()
            plan.run()
            prof = hal.prof_end()
            c1 = hal.group_counters()
        fused_ms, fused_n = prof["fold_eval_mfma"]
        # fused rounds of one prove on the group path: r = n_vars .. 2 (the first launch of a prove only evaluates)
        rec = {"bench": "claims", "n_vars": n_vars, "k": len(comps), "m": m, "kind": args.kind, "group": group, "chain_min_log2": os.environ.get("BN_GROUP_CHAIN_MIN_LOG2", "default"), "ms_per_prove": round(ms, 4),
               "whole_prove_frac_of_64mN": round(64.0 * m * n / (ms * 1e-3) / PEAK, 4), "verifier_check": bool(ok),
               "prof_ms": {kk: [round(v[0], 4), v[1]] for kk, v in prof.items() if v[1]},
               "group_counters_one_prove": {kk: c1[kk] - c0[kk] for kk in c1}}
        if group and fused_n:
            # algorithmic bytes of the fused launches of one prove: sum over rounds r = n_vars .. 2 of 24 * m * 2^r (arrays fused) --
            # with shared arrays the plain folds and the evaluation jobs move more, so the figure is quoted for what the prover asked for
            alg = sum(24.0 * m * (1 << r) for r in range(2, n_vars + 1))
            big = [r for r in range(2, n_vars + 1) if (1 << r) >= (1 << 19)]
            rec["launches_profiled"] = fused_n
            rec["algorithmic_GB_folds_plus_evals"] = round(alg / 1e9, 4)
            rec["group_launch_ms_total"] = round(fused_ms, 4)
            rec["group_launch_frac"] = round(alg / (fused_ms * 1e-3) / PEAK, 4) if fused_ms else None
        out.append(rec)
        print(json.dumps(rec))
        sys.stdout.flush()
    return out


def run_piop(args):
    import numpy as np

    import binius_amd
    from binius_amd import synthetic
    from binius_amd._host import FRIParams, PiopPlan

    n = args.n
    n_varss = [n - 3, n - 3, n - 1, n]
    total_elems = sum(1 << v for v in n_varss)
    total_vars = (total_elems - 1).bit_length()
    arities = []
    while sum(arities) + args.arity < total_vars:
        arities.append(args.arity)
    p = FRIParams(total_vars - args.log_batch, args.log_inv_rate, args.log_batch, arities, n_test_queries=3)
    sizes = sorted(set(n_varss))
    t_sizes = [v for v in sizes for _ in range(args.transparents)]
    for group in ([1, 0] if args.group < 0 else [args.group]):
        os.environ["BN_GROUP"] = str(group)
        code_elems = 1 << (total_vars + args.log_inv_rate)
        ml_elems = sum(1 << v for v in n_varss) + sum(1 << v for v in t_sizes)
        with binius_amd.Context(0, (1 << total_vars) + 4 * code_elems + 2 * ml_elems + (1 << 16)) as hal:
            F = binius_amd.HostField
            alloc = hal.dev_alloc()
            d_c = [(v, device_random(hal, alloc, F, 0x9100 + i, v)) for i, v in enumerate(n_varss)]
            d_t = [(v, device_random(hal, alloc, F, 0xA100 + j, v)) for j, v in enumerate(t_sizes)]
            # merge_multilins on the host (as the reference does): reversed order, bit-reversed indices
            msg = np.zeros((1 << total_vars, 2), dtype=np.uint64)
            at = 0
            for v, s in reversed(d_c):
                x = hal.copy_d2h(s)
                idx = np.arange(1 << v, dtype=np.int64)
                rev = np.zeros_like(idx)
                for b in range(v):
                    rev |= ((idx >> b) & 1) << (v - 1 - b)
                chunk = np.empty_like(x)
                chunk[rev] = x
                msg[at : at + (1 << v)] = chunk
                at += 1 << v
            d_msg = alloc.alloc(1 << total_vars)
            hal.copy_h2d(msg, d_msg)
            claims = []
            for i, (v, c) in enumerate(d_c):
                for j, (tv, t) in enumerate(d_t):
                    if v == tv:
                        claims.append((v, i, j, hal.inner_product(c, 7, t)))
            stream = synthetic.random_scalars(0x7A0 + n, len(sizes) + total_vars)
            bcs, chs = stream[: len(sizes)], stream[len(sizes) :]
            scratch = alloc.alloc(4 * code_elems + ml_elems + (1 << 14))
            plan = PiopPlan(hal, d_c, d_t, claims, p, d_msg, scratch, bcs, chs)
            for _ in range(args.warmup):
                plan.run()
            commit_ms, prove_ms = [], []
            for _ in range(args.steps):
                a, b = plan.run()
                commit_ms.append(a)
                prove_ms.append(b)
            c0 = hal.group_counters()
            hal.prof_begin()
            plan.run()
            prof = hal.prof_end()
            c1 = hal.group_counters()
        rec = {"bench": "piop", "n_varss": n_varss, "transparents_per_size": args.transparents, "claims": len(claims), "total_vars": total_vars,
               "fri": {"log_inv_rate": args.log_inv_rate, "log_batch": args.log_batch, "arities": arities}, "group": group,
               "commit_ms": round(min(commit_ms), 4), "prove_ms": round(min(prove_ms), 4), "prove_ms_mean": round(sum(prove_ms) / len(prove_ms), 4),
               "prof_ms": {kk: [round(v[0], 4), v[1]] for kk, v in prof.items() if v[1]},
               "group_counters_one_prove": {kk: c1[kk] - c0[kk] for kk in c1}}
        print(json.dumps(rec))
        sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("claims")
    a.add_argument("--n-vars", type=int, default=24)
    a.add_argument("--k", type=int, default=4)
    a.add_argument("--kind", default="disjoint")
    b = sub.add_parser("piop")
    b.add_argument("--n", type=int, default=20)
    b.add_argument("--log-inv-rate", type=int, default=1)
    b.add_argument("--log-batch", type=int, default=4)
    b.add_argument("--arity", type=int, default=4)
    b.add_argument("--transparents", type=int, default=2)
    for s in (a, b):
        s.add_argument("--steps", type=int, default=5)
        s.add_argument("--warmup", type=int, default=2)
        s.add_argument("--group", type=int, default=-1, help="1 / 0: the claim-group path on / off; -1: both")
    args = ap.parse_args()
    if args.cmd == "claims":
        run_claims(args)
    else:
        run_piop(args)


if __name__ == "__main__":
    main()
