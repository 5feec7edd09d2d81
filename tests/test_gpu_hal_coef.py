"""The old HAL's round evaluation in COEFFICIENT form (csrc/abi_hal.cpp round_evals_coef): compositions of degree
<= 3 over full Folded multilinears, High-to-Low, at X = 1, infinity and ANY number of interpolation-domain points -- the round
polynomial's coefficients come from bilinear sums over halves (one element-wise launch for the products a (.) b of the cubic
monomials, one launch of the claim groups' kernel for every sum), the domain points meet the 16-byte coefficients on the host.
Reference: crates/hal/src/sumcheck_round_calculation.rs:85-330 (every composition at every point of every vertex pair).
Every value against the oracle's restatement of CpuBackend (oracle/hal_ref.c), bit for bit; the same requests with the path
switched off (BN_HAL_COEF=0: rows + compiled circuits) give the same values."""
import os

import numpy as np
import pytest

from test_gpu_hal import upload

pytestmark = pytest.mark.gpu

AB = [("var", 0), ("var", 1), ("mul", 0, 1)]
ABC = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)]
ABC_PLUS_A = ABC + [("add", 4, 0)]
AAB = [("var", 0), ("var", 0), ("mul", 0, 1), ("var", 1), ("mul", 2, 3)]
K = 0x1234567890ABCDEF1122334455667788
# (a + K) * b * c + K' * a * b + c + K'': every degree, constants in the coefficients
MIXED = [("var", 0), ("const", K), ("add", 0, 1), ("var", 1), ("mul", 2, 3), ("var", 2), ("mul", 4, 5),
         ("const", K ^ 0x55), ("mul", 7, 0), ("mul", 8, 3), ("add", 6, 9), ("add", 10, 5), ("const", 77), ("add", 11, 12)]
MIXED_INF = ABC


def run(oracle, n_vars, n_mls, evaluators, points, env=None, expect_path=True, seed=0):
    """The request on a fresh context (env applied while it is created), the oracle's values beside it."""
    import binius_amd

    n = 1 << n_vars
    xs = [oracle.random_b128(0xC0EF0000 + 1024 * seed + 64 * n_vars + j, n) for j in range(n_mls)]
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        hal = binius_amd.Context(0, (n_mls + 2) * n + (1 << 16))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        alloc = hal.dev_alloc()
        d_mls = [("folded", upload(hal, alloc, v), 0) for v in xs]
        eqs = {}
        exprs, d_evs = [], []
        for e in evaluators:
            c, ci = hal.compile_expr(e["steps"]), hal.compile_expr(e["steps_inf"])
            exprs += [c, ci]
            d_eq = None
            if e.get("eq_ind") is not None:
                if id(e["eq_ind"]) not in eqs:
                    eqs[id(e["eq_ind"])] = upload(hal, alloc, e["eq_ind"])
                d_eq = eqs[id(e["eq_ind"])]
            d_evs.append({"composition": c, "composition_at_infinity": ci, "start": e["start"], "end": e["end"], "eq_ind": d_eq})
        before = hal.group_counters()["launches"]
        got = hal.hal_round_evals(1, n_vars, None, d_mls, d_evs, points)
        again = hal.hal_round_evals(1, n_vars, None, d_mls, d_evs, points)
        taken = hal.group_counters()["launches"] - before
        for x in exprs:
            x.free()
    finally:
        hal.close()
    rc, want = oracle.hal_round_evals(1, n_vars, None, [("folded", v, 0) for v in xs], evaluators, points)
    assert rc == 0
    assert got == want
    assert again == got
    if expect_path is not None:
        assert (taken == 2) == expect_path, "coefficient-form path %s (group launches: %d)" % ("not taken" if expect_path else "taken", taken)
    return got


@pytest.mark.parametrize("n_vars", [2, 3, 5, 9, 12, 15, 18, 20, 21])
@pytest.mark.parametrize("with_eq", [False, True])
def test_cubic_at_domain_points(oracle, n_vars, with_eq):
    """a * b * c + a at X = 1, infinity, z (tools/bench_hal.py's general request), with and without an indicator."""
    eq = oracle.random_b128(0xC0E1 + n_vars, (1 << n_vars) // 2) if with_eq else None
    pts = oracle.random_scalars(0xC0E2 + n_vars, 1)
    evs = [{"steps": ABC_PLUS_A, "steps_inf": ABC, "start": 1, "end": 4, "eq_ind": eq}]
    got = run(oracle, n_vars, 3, evs, pts)
    if n_vars in (5, 12):
        assert run(oracle, n_vars, 3, evs, pts, env={"BN_HAL_COEF": "0"}, expect_path=False) == got


@pytest.mark.parametrize("n_vars", [2, 4, 8, 13, 17])
@pytest.mark.parametrize("with_eq", [False, True])
def test_mixed_evaluators_and_point_ranges(oracle, n_vars, with_eq):
    """Several evaluators over five multilinears: every degree, repeated variables, constants, point ranges that start at 1, 2, 3
    and end at 3 ... 7 (four domain points), one evaluator with domain points only."""
    eq = oracle.random_b128(0xC0E3 + n_vars, (1 << n_vars) // 2) if with_eq else None
    pts = oracle.random_scalars(0xC0E4 + n_vars, 4)
    sub = lambda steps, m: [(("var", m[s[1]]) if s[0] == "var" else s) for s in steps]
    evs = [
        {"steps": sub(MIXED, [0, 1, 2]), "steps_inf": sub(MIXED_INF, [0, 1, 2]), "start": 1, "end": 7, "eq_ind": eq},
        {"steps": sub(AAB, [3, 4]), "steps_inf": sub(AAB, [3, 4]), "start": 2, "end": 5, "eq_ind": eq},
        {"steps": sub(AB, [1, 4]), "steps_inf": sub(AB, [1, 4]), "start": 1, "end": 4, "eq_ind": eq},
        {"steps": sub(ABC, [0, 1, 4]), "steps_inf": sub(ABC, [0, 1, 4]), "start": 3, "end": 6, "eq_ind": eq},
        {"steps": [("var", 2), ("const", 5), ("add", 0, 1)], "steps_inf": [("var", 2)], "start": 1, "end": 5, "eq_ind": eq},
        {"steps": sub(ABC_PLUS_A, [2, 3, 0]), "steps_inf": sub(ABC, [2, 3, 0]), "start": 1, "end": 3, "eq_ind": eq},
    ]
    run(oracle, n_vars, 5, evs, pts, seed=1)


@pytest.mark.parametrize("n_vars", [3, 10, 16])
def test_cubic_at_one_and_infinity_only(oracle, n_vars):
    """Degree 3 without domain points (a * b * c, a^2 * b at X = 1, infinity; the second at infinity only): two products per triple."""
    evs = [{"steps": ABC, "steps_inf": ABC, "start": 1, "end": 3, "eq_ind": None},
           {"steps": AAB, "steps_inf": AAB, "start": 2, "end": 3, "eq_ind": None}]
    run(oracle, n_vars, 3, evs, [])


@pytest.mark.parametrize("n_vars", [6, 14])
def test_degree_two_without_domain_points_keeps_the_routed_code(oracle, n_vars):
    evs = [{"steps": AB, "steps_inf": AB, "start": 1, "end": 3, "eq_ind": None}]
    run(oracle, n_vars, 2, evs, [], expect_path=False)


@pytest.mark.parametrize("n_vars", [4, 11, 15])
@pytest.mark.parametrize("with_eq", [False, True])
def test_wide_cubic_request(oracle, n_vars, with_eq):
    """Forty multilinears, thirty evaluators of degree 3 (random triples, shared pairs) with two domain points: beyond what one pass
    of the general code carries (16 multilinears / 8 evaluators) -- one request here."""
    rng = np.random.RandomState(0xC0E5 + n_vars)
    eq = oracle.random_b128(0xC0E6 + n_vars, (1 << n_vars) // 2) if with_eq else None
    pts = oracle.random_scalars(0xC0E7 + n_vars, 2)
    evs = []
    for i in range(30):
        a, b, c, d = (int(x) for x in rng.randint(0, 40, 4))
        if i % 3 == 0:
            a, b = 7, 11  # (a shared product pair)
        steps = [("var", a), ("var", b), ("mul", 0, 1), ("var", c), ("mul", 2, 3), ("var", d), ("add", 4, 5)]
        inf = [("var", a), ("var", b), ("mul", 0, 1), ("var", c), ("mul", 2, 3)]
        evs.append({"steps": steps, "steps_inf": inf, "start": 1, "end": 5, "eq_ind": eq})
    run(oracle, n_vars, 40, evs, pts, seed=2)


def test_truncated_or_low_to_high_requests_keep_the_general_code(oracle):
    """A truncated multilinear / an evaluation point 0 declines the path (the general code answers, same values as the oracle)."""
    import binius_amd

    n_vars, n = 8, 256
    xs = [oracle.random_b128(0xC0E8 + j, n) for j in range(3)]
    pts = oracle.random_scalars(0xC0E9, 1)
    with binius_amd.Context(0, 8 * n + (1 << 16)) as hal:
        alloc = hal.dev_alloc()
        c, ci = hal.compile_expr(ABC_PLUS_A), hal.compile_expr(ABC)
        for mls, start in (([("folded", xs[0], 0), ("folded", xs[1][: n - 5], 9), ("folded", xs[2], 0)], 1), ([("folded", v, 0) for v in xs], 0)):
            d_mls = [(m[0], upload(hal, alloc, m[1])) + tuple(m[2:]) for m in mls]
            before = hal.group_counters()["launches"]
            got = hal.hal_round_evals(1, n_vars, None, d_mls, [{"composition": c, "composition_at_infinity": ci, "start": start, "end": 4, "eq_ind": None}], pts)
            assert hal.group_counters()["launches"] == before
            rc, want = oracle.hal_round_evals(1, n_vars, None, mls, [{"steps": ABC_PLUS_A, "steps_inf": ABC, "start": start, "end": 4, "eq_ind": None}], pts)
            assert rc == 0 and got == want
        c.free()
        ci.free()
