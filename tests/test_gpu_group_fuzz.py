"""Differential fuzz of the claim-group path at the sizes where the dispatcher switches kernels: one or two
BivariateSumcheckProver-shaped call sequences on ONE context in the reference's order -- execute on every prover, one challenge,
fold on every prover (front_loaded.rs:122-155; the first fold of a prover copies evals_0 into a fresh buffer,
v3/bivariate_product.rs:196-206) -- with k in {1, 2, 3} product claims over up to 6 multilinears per prover (claims sharing a
multilinear, a multilinear in no claim, the occasional square), batch coefficients != 1, arrays of 2^16 .. 2^21 elements, and
foreign calls between any two steps: reads of a prover's arrays (copy_d2h), copies out of and into them, fills and kernels on
unrelated memory, fri_fold on a codeword beside them, synchronisations, a prover finishing early (the reads of finish()).

Three-way: the deferred execution (claim groups on) == eager execution (BN_NO_LAZY_FOLD=1) == a host model driven by the
oracle -- every returned value, and at the end every byte of every buffer."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))


@pytest.mark.parametrize("seed", list(range(28)))
def test_group_call_sequences_three_way(monkeypatch, oracle, seed):
    import binius_amd
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    rng = np.random.RandomState(0x6F0 + 7919 * seed)
    n_provers = int(rng.choice([1, 2, 2]))
    log_top = int(rng.randint(16, 22)) if seed < 16 else int(rng.randint(6, 15))  # (the small ones: hosted sessions from early on)
    specs = []
    for p in range(n_provers):
        log_n = log_top - (int(rng.randint(0, 4)) if p + 1 < n_provers else 0)
        m = int(rng.randint(2, 7))
        k = int(rng.choice([1, 2, 3]))
        comps = []
        for c in range(k):
            i, j = int(rng.randint(m)), int(rng.randint(m))
            if i == j and rng.rand() < 0.8:
                j = (i + 1) % m
            comps.append((i, j))
        specs.append((log_n, m, comps))
    specs.sort(key=lambda s: s[0])
    total = sum(m << log_n for log_n, m, _ in specs)
    spare_n = 1 << 14
    arena = total + total // 2 + 4 * spare_n + (1 << 12)
    monkeypatch.setenv("BN_NO_LAZY_FOLD", "1")
    eager = binius_amd.Context(0, arena)
    monkeypatch.delenv("BN_NO_LAZY_FOLD")
    if seed % 2:  # (chains -- jobs of one launch that read what other jobs of it fold -- at every size; the default: from 2^21 points)
        monkeypatch.setenv("BN_GROUP_CHAIN_MIN_LOG2", "0")
    lazy = binius_amd.Context(0, arena)
    monkeypatch.delenv("BN_GROUP_CHAIN_MIN_LOG2", raising=False)
    threads = _threads()
    s_evals = oracle.ntt_s_evals(5, 12)
    try:
        # ---- host model
        model = [[oracle.random_b128(0x6F000000 + 4096 * seed + 64 * p + j, 1 << log_n) for j in range(m)] for p, (log_n, m, _) in enumerate(specs)]
        folded_model = [[None] * m for _, m, _ in specs]  # the "fresh buffer" of each multilinear after its first fold
        spare_model = [oracle.random_b128(0x6F100000 + 16 * seed + j, spare_n) for j in range(3)]
        ctxs = []
        for hal in (eager, lazy):
            alloc = hal.dev_alloc()
            inputs = []
            for p, (log_n, m, _) in enumerate(specs):
                row = []
                for j in range(m):
                    d = alloc.alloc(1 << log_n)
                    hal.copy_h2d(model[p][j], d)
                    row.append(d)
                inputs.append(row)
            fresh = [[alloc.alloc(1 << (log_n - 1)) for _ in range(m)] for log_n, m, _ in specs]
            spare = [alloc.alloc(spare_n) for _ in range(3)]
            for s, h in zip(spare, spare_model):
                hal.copy_h2d(h, s)
            cw_out = alloc.alloc(spare_n >> 2)
            exprs = [[bivariate_product_expr(hal, i, j) for i, j in comps] for _, _, comps in specs]
            ctxs.append(dict(hal=hal, inputs=inputs, fresh=fresh, spare=spare, cw_out=cw_out, exprs=exprs))
        cw_model = oracle.arr(spare_n >> 2)
        zs = iter(oracle.random_scalars(0x6F2 + seed, 512))
        bcs = [next(zs) for _ in specs]
        n_rem = [log_n for log_n, _, _ in specs]  # variables left per prover
        pre_fold = [True] * len(specs)
        alive = [True] * len(specs)

        def cur_model(p, j):
            """the live array of multilinear j of prover p"""
            return (model[p][j] if pre_fold[p] else folded_model[p][j])[: 1 << n_rem[p]]

        def cur_dev(c, p, j):
            return (c["inputs"][p][j] if pre_fold[p] else c["fresh"][p][j]).slice(0, 1 << n_rem[p])

        def same(outs, want, what):
            assert outs[0] == outs[1], ("deferred != eager", what)
            if want is not None:
                assert outs[1] == want, ("device != oracle", what)

        def execute(p):
            log_n, m, comps = specs[p]
            coeffs, pw = [], 1
            for _ in comps:
                coeffs.append(pw)
                pw = oracle.mul(pw, bcs[p])
            outs = [calculate_round_evals(c["hal"], n_rem[p], coeffs, [cur_dev(c, p, j) for j in range(m)], c["exprs"][p]) for c in ctxs]
            rc, want = oracle.round_evals([cur_model(p, j).copy() for j in range(m)], n_rem[p], comps, bcs[p], threads=threads)
            assert rc == 0
            same(outs, want, ("execute", p, n_rem[p]))

        def fold(p, z):
            log_n, m, comps = specs[p]
            half = 1 << (n_rem[p] - 1)
            for c in ctxs:
                hal = c["hal"]
                if pre_fold[p]:
                    for j in range(m):  # allocate a new buffer for the folded evaluations and copy in evals_0
                        hal.copy_d2d(c["inputs"][p][j].slice(0, half), c["fresh"][p][j].slice(0, half))
                    e0 = [c["fresh"][p][j].slice(0, half) for j in range(m)]
                    e1 = [c["inputs"][p][j].slice(half, 2 * half) for j in range(m)]
                else:
                    e0 = [c["fresh"][p][j].slice(0, half) for j in range(m)]
                    e1 = [c["fresh"][p][j].slice(half, 2 * half) for j in range(m)]
                hal.extrapolate_line_batch(e0, e1, z)
            for j in range(m):
                src = cur_model(p, j)
                f = src[:half].copy()
                assert oracle.extrapolate_line(f, src[half : 2 * half].copy(), z) == 0
                if pre_fold[p]:
                    folded_model[p][j] = np.zeros((1 << (log_n - 1), 2), dtype=np.uint64)
                folded_model[p][j][:half] = f
            pre_fold[p] = False
            n_rem[p] -= 1

        def foreign():
            kind = str(rng.choice(["read", "read_spare", "copy_out", "copy_in", "fill", "kernel", "fri_fold", "sync"]))
            live = [p for p in range(len(specs)) if alive[p]]
            if not live:
                return
            p = int(rng.choice(live))
            j = int(rng.randint(specs[p][1]))
            if kind == "read":
                w = min(4, 1 << n_rem[p])
                outs = [c["hal"].copy_d2h(cur_dev(c, p, j).slice(0, w)).tolist() for c in ctxs]
                same(outs, cur_model(p, j)[:w].tolist(), "read")
            elif kind == "read_spare":
                outs = [c["hal"].copy_d2h(c["spare"][0].slice(0, 8)).tolist() for c in ctxs]
                same(outs, spare_model[0][:8].tolist(), "read_spare")
            elif kind == "copy_out":
                w = min(spare_n, 1 << n_rem[p])
                for c in ctxs:
                    c["hal"].copy_d2d(cur_dev(c, p, j).slice(0, w), c["spare"][1].slice(0, w))
                spare_model[1][:w] = cur_model(p, j)[:w]
            elif kind == "copy_in":
                if pre_fold[p]:
                    return  # (PreFold inputs are read-only for the prover; a test must not write them either)
                w = min(64, 1 << n_rem[p])
                for c in ctxs:
                    c["hal"].copy_d2d(c["spare"][0].slice(0, w), cur_dev(c, p, j).slice(0, w))
                folded_model[p][j][:w] = spare_model[0][:w]
            elif kind == "fill":
                z = next(zs)
                for c in ctxs:
                    c["hal"].fill(c["spare"][2].slice(0, 1024), z)
                spare_model[2][:1024] = (z & ((1 << 64) - 1), z >> 64)
            elif kind == "kernel":
                outs = [c["hal"].inner_product(c["spare"][0], 7, c["spare"][1]) for c in ctxs]
                rc, want = oracle.inner_product(spare_model[0], 7, spare_model[1])
                same(outs, want, "kernel")
            elif kind == "fri_fold":
                chs = [next(zs), next(zs)]
                for c in ctxs:
                    c["hal"].fri_fold(s_evals, 5, 12, 12, 2, chs, c["spare"][0], c["cw_out"])
                assert oracle.fri_fold(s_evals, 5, 12, 12, 2, chs, spare_model[0], cw_model) == 0
            else:
                for c in ctxs:
                    c["hal"].sync()

        def finish(p):
            m = specs[p][1]
            for j in range(m):
                outs = [c["hal"].copy_d2h(cur_dev(c, p, j).slice(0, 1)).tolist() for c in ctxs]
                same(outs, cur_model(p, j)[:1].tolist(), ("finish", p, j))
            alive[p] = False

        n_rounds = int(rng.randint(4, 9))
        for rnd in range(n_rounds):
            for p in range(len(specs)):
                if alive[p] and n_rem[p] >= 1:
                    if rng.rand() < 0.12:
                        foreign()
                    execute(p)
            z = next(zs)
            if rng.rand() < 0.1:
                foreign()
            for p in range(len(specs)):
                if alive[p] and n_rem[p] >= 1:
                    fold(p, z)
                    if rng.rand() < 0.1:
                        foreign()
            if rng.rand() < 0.08 and sum(alive) > 1:
                finish(int(rng.choice([p for p in range(len(specs)) if alive[p]])))
            if rng.rand() < 0.25:
                foreign()
        for p in range(len(specs)):
            if alive[p] and n_rem[p] >= 1 and rng.rand() < 0.5:
                execute(p)  # (a sequence may end on an evaluation or on a fold)
        # ---- every byte of every buffer
        for p, (log_n, m, _) in enumerate(specs):
            for j in range(m):
                a, b = (c["hal"].copy_d2h(c["inputs"][p][j]) for c in ctxs)
                assert np.array_equal(a, b) and np.array_equal(b, model[p][j]), ("input", p, j)
                if folded_model[p][j] is not None:
                    live = 1 << n_rem[p]
                    a, b = (c["hal"].copy_d2h(c["fresh"][p][j]) for c in ctxs)
                    assert np.array_equal(a, b), ("folded buffer: deferred != eager", p, j)
                    assert np.array_equal(b[:live], folded_model[p][j][:live]), ("folded buffer != model", p, j)
        for q in range(3):
            a, b = (c["hal"].copy_d2h(c["spare"][q]) for c in ctxs)
            assert np.array_equal(a, b) and np.array_equal(b, spare_model[q]), ("spare", q)
        cnt = lazy.group_counters()
        if n_provers > 1 or any(len(set(c)) > 1 or m > 2 for _, m, c in specs):
            assert cnt["evals"] > 0, cnt
    finally:
        eager.close()
        lazy.close()
