"""Differential test of the ABI's deferred execution (pending folds and copies, fold + evaluation
fusion, host mirror of tiny folds, resident tail) against eager execution (BN_NO_LAZY_FOLD=1): random
sequences of ComputeLayer calls must leave identical device memory and return identical values."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _contexts(monkeypatch, tail):
    import binius_amd

    monkeypatch.setenv("BN_NO_LAZY_FOLD", "1")
    eager = binius_amd.Context(0, 1 << 16)
    monkeypatch.delenv("BN_NO_LAZY_FOLD")
    if tail:
        monkeypatch.setenv("BN_TAIL_MAX_LOG2", "12")
    lazy = binius_amd.Context(0, 1 << 16)
    monkeypatch.delenv("BN_TAIL_MAX_LOG2", raising=False)
    return eager, lazy


@pytest.mark.parametrize("seed", list(range(12)))
@pytest.mark.parametrize("tail", [False, True])
def test_random_call_sequences(monkeypatch, oracle, seed, tail):
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    eager, lazy = _contexts(monkeypatch, tail)
    try:
        rng = np.random.RandomState(1000 * seed + (7 if tail else 0))
        log_n = int(rng.randint(4, 11))
        n = 1 << log_n
        n_bufs = 6
        host = [oracle.random_b128(0xFA220000 + 16 * seed + j, n) for j in range(n_bufs)]
        ctxs = []
        for hal in (eager, lazy):
            alloc = hal.dev_alloc()
            bufs = [alloc.alloc(n) for _ in range(n_bufs)]
            for b, h in zip(bufs, host):
                hal.copy_h2d(h, b)
            hal._nodes = alloc.alloc(4 * n)  # room for the Merkle tree of any live prefix
            ctxs.append((hal, bufs, bivariate_product_expr(hal, 0, 1)))
        zs = oracle.random_scalars(0xFA22 + seed, 64)
        cur = n  # current "live" length of the arrays (they shrink when folded)
        log = []
        for step in range(40):
            op = rng.choice(["fold2", "fold_any", "copy", "eval", "read", "fill", "fold_copy_first", "chain", "merkle", "gather"])
            a, b, c = [int(x) for x in rng.choice(n_bufs, 3, replace=False)]
            z = zs[step]
            outs = []
            for hal, bufs, expr in ctxs:
                half = cur // 2
                if op == "fold2" and cur >= 2:
                    # the prover's shape: two arrays folded in place, then (usually) evaluated
                    e0 = [bufs[a].slice(0, half), bufs[b].slice(0, half)]
                    e1 = [bufs[a].slice(half, cur), bufs[b].slice(half, cur)]
                    hal.extrapolate_line_batch(e0, e1, z)
                    if half >= 2 and step % 3 != 2:
                        outs.append(calculate_round_evals(hal, int(np.log2(half)), [1], [bufs[a].slice(0, half), bufs[b].slice(0, half)], [expr]))
                elif op == "fold_any" and cur >= 2:
                    k = int(rng.randint(1, 4)) if hal is ctxs[0][0] else None
                    # same k on both contexts: draw it once
                    if hal is ctxs[0][0]:
                        log.append(k)
                    else:
                        k = log[-1]
                    idx = [a, b, c][:k]
                    hal.extrapolate_line_batch([bufs[i].slice(0, half) for i in idx], [bufs[i].slice(half, cur) for i in idx], z)
                elif op == "copy":
                    hal.copy_d2d(bufs[a].slice(0, cur), bufs[b].slice(0, cur))
                elif op == "eval" and cur >= 2:
                    outs.append(calculate_round_evals(hal, int(np.log2(cur)), [1], [bufs[a].slice(0, cur), bufs[b].slice(0, cur)], [expr]))
                elif op == "read":
                    outs.append(hal.copy_d2h(bufs[a].slice(0, min(cur, 4))).tolist())
                elif op == "fill":
                    hal.fill(bufs[c].slice(0, cur), z)
                elif op == "merkle":
                    # the commitment entry points observe device memory: deferred folds / copies must land first
                    batch = 2 if cur >= 4 else 1
                    nodes = hal._nodes.slice(0, 2 * (2 * (cur // batch) - 1))
                    hal.merkle_build(bufs[a].slice(0, cur), batch, nodes)
                    outs.append(hal.copy_d2h(nodes.slice(nodes.len - 2, nodes.len)).tolist())
                elif op == "gather":
                    outs.append(hal.gather_d2h(bufs[b].slice(0, cur), [0, cur - 1, cur // 2], 1).tolist())
                elif op == "fold_copy_first" and cur >= 2:
                    # first fold of a prover: copy evals_0 into fresh buffers, fold those
                    hal.copy_d2d(bufs[a].slice(0, half), bufs[c].slice(0, half))
                    hal.extrapolate_line_batch([bufs[c].slice(0, half)], [bufs[a].slice(half, cur)], z)
                elif op == "chain" and cur >= 2:
                    # copy chain feeding a fold of both destinations (must not be absorbed blindly)
                    hal.copy_d2d(bufs[a].slice(0, half), bufs[b].slice(0, half))
                    hal.copy_d2d(bufs[b].slice(0, half), bufs[c].slice(0, half))
                    hal.extrapolate_line_batch([bufs[b].slice(0, half), bufs[c].slice(0, half)],
                                               [bufs[a].slice(half, cur), bufs[a].slice(half, cur)], z)
            if op in ("fold2",) and cur >= 4 and rng.rand() < 0.5:
                cur //= 2
            assert len(outs) in (0, 2) and (not outs or outs[0] == outs[1]), (step, op)
        for j in range(n_bufs):
            assert np.array_equal(ctxs[0][0].copy_d2h(ctxs[0][1][j]), ctxs[1][0].copy_d2h(ctxs[1][1][j])), j
    finally:
        eager.close()
        lazy.close()
