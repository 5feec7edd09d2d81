"""The old HAL's round evaluation at the width of a constraint set's zerocheck (VERDICT r5 missing 2 / item 5): the reference hands
`sumcheck_compute_round_evals` ONE evaluator per constraint over EVERY multilinear of the table
(core/src/constraint_system/prove.rs:431-505, hal/src/backend.rs:52-67) -- for the keccak table a hundred compositions
(chi: out - (b0 + (b1 - 1) b2), chi + iota with the round constant, the link rule (out - next_in) * sel;
m3/src/gadgets/hash/keccak/stacked.rs:142-151,340-363) over two hundred columns, each composition reading three to five of them.
bn_hal_round_evals deals such a request out to passes of at most 8 evaluators over at most 16 multilinears (csrc/abi_hal.cpp
round_evals_in_parts); every value against the oracle's restatement of CpuBackend (oracle/hal_ref.c), bit for bit."""
import numpy as np
import pytest

from test_gpu_hal import upload

pytestmark = pytest.mark.gpu


def keccak_constraints(n_batches=3):
    """(n_multilinears, evaluators-as-steps) of the keccak table's constraint set in the variable numbering of the zerocheck
    prover: per batch 25 state_out, 25 b, 1 round constant; then 25 packed state_out, 25 next_state_in, 1 selector."""
    evs = []
    n = 0
    for _ in range(n_batches):
        out0, b0, rc = n, n + 25, n + 50
        n += 51
        for x in range(5):
            for y in range(5):
                o, bb0, bb1, bb2 = out0 + 5 * y + x, b0 + 5 * y + x, b0 + 5 * y + (x + 1) % 5, b0 + 5 * y + (x + 2) % 5
                # output - (rc? + b0 + (b1 - 1) * b2); characteristic 2: minus is plus
                steps = [("var", bb1), ("const", 1), ("add", 0, 1), ("var", bb2), ("mul", 2, 3), ("var", bb0), ("add", 4, 5), ("var", o), ("add", 6, 7)]
                if (x, y) == (0, 0):
                    steps += [("var", rc), ("add", 8, 9)]
                evs.append((steps, [("var", bb1), ("var", bb2), ("mul", 0, 1)]))
    sop, nsi, sel = n, n + 25, n + 50
    n += 51
    for i in range(25):
        prod = [("var", sop + i), ("var", nsi + i), ("add", 0, 1), ("var", sel), ("mul", 2, 3)]
        evs.append((prod, prod))
    return n, evs


@pytest.mark.parametrize("n_vars,n_batches", [(4, 3), (10, 3), (14, 1), (17, 1)])
def test_keccak_constraint_set_vs_oracle(oracle, n_vars, n_batches):
    """One call with every constraint of the (1- or 3-batch) keccak table as an equality-indicator evaluator at X = 1, infinity,
    High-to-Low, all multilinears full: 50 / 100 evaluators over 102 / 204 multilinears."""
    import binius_amd

    n_mls, cons = keccak_constraints(n_batches)
    n = 1 << n_vars
    xs = [oracle.random_b128(0x6EC0000 + 64 * n_vars + j, n) for j in range(n_mls)]
    eq = oracle.random_b128(0x6EC1 + n_vars, n // 2)
    evaluators = [{"steps": s, "steps_inf": si, "start": 1, "end": 3, "eq_ind": eq} for s, si in cons]
    mls = [("folded", v, 0) for v in xs]
    with binius_amd.Context(0, (n_mls + 2) * n + (1 << 16)) as hal:
        alloc = hal.dev_alloc()
        d_mls = [("folded", upload(hal, alloc, v), 0) for v in xs]
        d_eq = upload(hal, alloc, eq)
        exprs, d_evs = [], []
        for s, si in cons:
            c, ci = hal.compile_expr(s), hal.compile_expr(si)
            exprs += [c, ci]
            d_evs.append({"composition": c, "composition_at_infinity": ci, "start": 1, "end": 3, "eq_ind": d_eq})
        got = hal.hal_round_evals(1, n_vars, None, d_mls, d_evs, [])
        again = hal.hal_round_evals(1, n_vars, None, d_mls, d_evs, [])
        for x in exprs:
            x.free()
    rc, want = oracle.hal_round_evals(1, n_vars, None, mls, evaluators, [])
    assert rc == 0
    assert got == want
    assert again == got


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("n_vars", [3, 9, 13])
def test_wide_mixed_request_vs_oracle(oracle, order, n_vars):
    """Forty multilinears (full, truncated with a suffix, Transparent at the tensor query) and twenty-two evaluators with
    different compositions, point ranges (0, 1, infinity, interpolation points) and indicators, in both orders: the parts the
    request is dealt out to give what the whole gives."""
    import binius_amd

    rng = np.random.RandomState(0x4A1D + n_vars + 7 * order)
    n = 1 << n_vars
    n_mls, n_evs = 40, 22
    query_vars = 2
    query = oracle.random_b128(0x4A1E, 1 << query_vars)
    mls = []
    for j in range(n_mls):
        kind = rng.randint(0, 8)
        if kind == 0 and n_vars >= 3:
            mls.append(("folded", np.ascontiguousarray(oracle.random_b128(0x4A20 + j, n)[: n - int(rng.randint(1, n // 2))]), oracle.random_scalars(0x4A30 + j, 1)[0]))
        elif kind == 1:
            level = int(rng.choice([0, 3, 5, 7]))
            mls.append(("transparent", oracle.random_b128(0x4A40 + j, max(1, (n << query_vars) >> (7 - level))), level, n_vars + query_vars))
        else:
            mls.append(("folded", oracle.random_b128(0x4A50 + j, n), 0))
    mls = [m for m in mls if not (m[0] == "transparent" and (m[1].shape[0] << (7 - m[2])) != (n << query_vars))] or mls
    n_mls = len(mls)
    eq = oracle.random_b128(0x4A60, max(1, n // 2))
    pts = oracle.random_scalars(0x4A61, 3)
    evaluators = []
    for e in range(n_evs):
        a, b, c = (int(v) for v in rng.choice(n_mls, 3, replace=False))
        form = e % 4
        if form == 0:
            steps, inf = [("var", a), ("var", b), ("mul", 0, 1), ("var", c), ("add", 2, 3)], [("var", a), ("var", b), ("mul", 0, 1)]
        elif form == 1:
            steps = [("var", a), ("var", b), ("mul", 0, 1), ("var", c), ("mul", 2, 3), ("var", a), ("add", 4, 5)]
            inf = [("var", a), ("var", b), ("mul", 0, 1), ("var", c), ("mul", 2, 3)]
        elif form == 2:
            steps = inf = [("var", a), ("var", b), ("mul", 0, 1)]
        else:
            steps = [("var", a), ("pow", 0, 3), ("const", 0x1234567890ABCDEF1122334455667788), ("mul", 1, 2), ("var", b), ("add", 3, 4)]
            inf = [("var", a), ("pow", 0, 3), ("const", 0x1234567890ABCDEF1122334455667788), ("mul", 1, 2)]
        start = int(rng.randint(0, 3))
        end = 6 if e == 0 else int(rng.randint(start + 1, 7))  # (one evaluator reaches the last interpolation point: three of them are passed)
        evaluators.append({"steps": steps, "steps_inf": inf, "start": start, "end": end, "eq_ind": eq if e % 3 == 0 else None})
    with binius_amd.Context(0, (n_mls + 4) * (n << query_vars) + (1 << 18)) as hal:
        alloc = hal.dev_alloc()
        d_mls = [(m[0], upload(hal, alloc, m[1])) + tuple(m[2:]) for m in mls]
        d_q = upload(hal, alloc, query)
        d_eq = upload(hal, alloc, eq)
        exprs, d_evs = [], []
        for e in evaluators:
            c, ci = hal.compile_expr(e["steps"]), hal.compile_expr(e["steps_inf"])
            exprs += [c, ci]
            d_evs.append({"composition": c, "composition_at_infinity": ci, "start": e["start"], "end": e["end"], "eq_ind": d_eq if e["eq_ind"] is not None else None})
        got = hal.hal_round_evals(order, n_vars, d_q, d_mls, d_evs, pts)
        for x in exprs:
            x.free()
    rc, want = oracle.hal_round_evals(order, n_vars, query, mls, evaluators, pts)
    assert rc == 0
    assert got == want
