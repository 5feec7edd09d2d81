"""The old HAL entry points (bn_hal_round_evals / bn_hal_fold_multilinear: binius_hal::ComputationBackend,
crates/hal/src/backend.rs:52-78) through the C ABI against the oracle's restatement of CpuBackend
(oracle/hal_ref.c, pinned by tests/test_oracle_hal.py).  Bit-exact.

Covered: both evaluation orders; Folded multilinears full, truncated with a constant suffix and empty;
Transparent multilinears of every tower level partially evaluated at the tensor query; regular and
equality-indicator evaluators with different evaluation point ranges (0, 1, infinity, interpolation-domain
points); the routed fast shape (products of two full multilinears at X = 1 / infinity, High-to-Low: 9-lane
kernels, and from 2^17 points the matrix-core kernels); the switchover fold; the reference's error cases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AB = [("var", 0), ("var", 1), ("mul", 0, 1)]
AB_PLUS_C = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3)]
ABC_PLUS_A = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3), ("add", 4, 0)]
ABC = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)]
SQ_PLUS = [("var", 1), ("pow", 0, 3), ("const", 0x1234567890ABCDEF1122334455667788), ("mul", 1, 2), ("var", 0), ("add", 3, 4)]
SQ_INF = [("var", 1), ("pow", 0, 3), ("const", 0x1234567890ABCDEF1122334455667788), ("mul", 1, 2)]


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, (1 << 22) + (1 << 20))
    yield ctx
    ctx.close()


def upload(hal, alloc, arr):
    d = alloc.alloc(max(1, arr.shape[0]))
    if arr.shape[0]:
        hal.copy_h2d(arr, d.slice(0, arr.shape[0]))
    return d.slice(0, arr.shape[0])


def both(hal, oracle, order, n_vars, query, mls, evaluators, points):
    """mls: ('folded', array, suffix) | ('transparent', array, level, n_vars_ml); evaluators: oracle-style dicts."""
    alloc = hal.dev_alloc()
    d_mls = [(m[0], upload(hal, alloc, m[1])) + tuple(m[2:]) for m in mls]
    d_q = upload(hal, alloc, query) if query is not None else None
    exprs, d_evs = [], []
    for e in evaluators:
        c, ci = hal.compile_expr(e["steps"]), hal.compile_expr(e["steps_inf"])
        exprs += [c, ci]
        d_evs.append({"composition": c, "composition_at_infinity": ci, "start": e["start"], "end": e["end"],
                      "eq_ind": upload(hal, alloc, e["eq_ind"]) if e.get("eq_ind") is not None else None})
    try:
        got = hal.hal_round_evals(order, n_vars, d_q, d_mls, d_evs, points)
    finally:
        for x in exprs:
            x.free()
    rc, want = oracle.hal_round_evals(order, n_vars, query, mls, evaluators, points)
    assert rc == 0
    return got, want


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("n_vars", [1, 2, 3, 7, 10, 13])
def test_round_evals_general(hal, oracle, order, n_vars):
    n = 1 << n_vars
    x = [oracle.random_b128(0x11A0 + 16 * n_vars + j, n) for j in range(3)]
    lens = [n, max(1, n - 3), n // 2]
    sfx = oracle.random_scalars(0x11B0, 3)
    eq = oracle.random_b128(0x11C0, max(1, n // 2))
    pts = oracle.random_scalars(0x11D0, 3)
    evaluators = [
        {"steps": AB_PLUS_C, "steps_inf": AB, "start": 0, "end": 4, "eq_ind": None},
        {"steps": ABC_PLUS_A, "steps_inf": ABC, "start": 1, "end": 6, "eq_ind": eq},
        {"steps": AB, "steps_inf": AB, "start": 2, "end": 3, "eq_ind": None},
        {"steps": SQ_PLUS, "steps_inf": SQ_INF, "start": 3, "end": 6, "eq_ind": None},
    ]
    mls = [("folded", np.ascontiguousarray(x[j][: lens[j]]), sfx[j] if j else 0) for j in range(3)]
    got, want = both(hal, oracle, order, n_vars, None, mls, evaluators, pts)
    assert got == want


@pytest.mark.parametrize("n_vars", [2, 6, 12, 16, 18, 19])
@pytest.mark.parametrize("with_eq", [False, True])
def test_round_evals_fast_shape(hal, oracle, n_vars, with_eq):
    """Every evaluator a * b at X = 1, infinity over full multilinears, High-to-Low: the ComputeLayer kernels
    (from n_vars = 18 the 2^17-point halves run on the matrix cores when there is no equality indicator)."""
    n = 1 << n_vars
    x = [oracle.random_b128(0x12A0 + j, n) for j in range(3)]
    eq = oracle.random_b128(0x12C0, n // 2) if with_eq else None
    evaluators = [{"steps": [("var", a), ("var", b), ("mul", 0, 1)], "steps_inf": [("var", a), ("var", b), ("mul", 0, 1)], "start": s, "end": 3, "eq_ind": eq}
                  for a, b, s in ((0, 1, 1), (2, 0, 1), (1, 2, 2))]
    got, want = both(hal, oracle, 1, n_vars, None, [("folded", v, 0) for v in x], evaluators, [])
    assert got == want
    # the same request through the general kernel (Low-to-High of the interleaved arrays) gives the same sums
    if n_vars <= 16:
        inter = []
        for v in x:
            y = np.empty_like(v)
            y[0::2], y[1::2] = v[: n // 2], v[n // 2 :]
            inter.append(y)
        got2, _ = both(hal, oracle, 0, n_vars, None, [("folded", v, 0) for v in inter], evaluators, [])
        assert got2 == got


@pytest.mark.parametrize("n_vars", [3, 9, 14])
@pytest.mark.parametrize("with_eq", [False, True])
def test_round_evals_fast_shape_three_factors(hal, oracle, n_vars, with_eq):
    """Products of three full multilinears (a * b * c, a^2 * b), with or without the equality indicator: routed to the
    bit-sliced product-sum kernel."""
    n = 1 << n_vars
    x = [oracle.random_b128(0x12E0 + j, n) for j in range(3)]
    eq = oracle.random_b128(0x12F0, n // 2) if with_eq else None
    abc = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)]
    aab = [("var", 0), ("var", 0), ("mul", 0, 1), ("var", 1), ("mul", 2, 3)]
    evaluators = [{"steps": abc, "steps_inf": abc, "start": 1, "end": 3, "eq_ind": eq},
                  {"steps": aab, "steps_inf": aab, "start": 2, "end": 3, "eq_ind": eq}]
    got, want = both(hal, oracle, 1, n_vars, None, [("folded", v, 0) for v in x], evaluators, [])
    assert got == want


@pytest.mark.parametrize("n_vars", [2, 5, 11, 18, 19])
def test_round_evals_routed_sums_of_products(hal, oracle, n_vars):
    """Compositions that are sums of monomials (a * b + c, a * b * c + a, K * b^3 + a, a * b + K, a single variable),
    with and without an equality indicator, at X = 1 and infinity over full multilinears: one product-sum pass per
    distinct monomial, coefficients applied on the host.  (Zerocheck-style constraints take this route.)"""
    n = 1 << n_vars
    x = [oracle.random_b128(0x18A0 + j, n) for j in range(3)]
    eq = oracle.random_b128(0x18B0, n // 2)
    K = 0x0123456789ABCDEF0FEDCBA987654321
    ab_plus_k = [("var", 0), ("var", 1), ("mul", 0, 1), ("const", K), ("add", 2, 3)]
    lin = [("var", 2)]
    kab = [("const", K), ("var", 0), ("mul", 0, 1), ("var", 1), ("mul", 2, 3)]  # K * a * b
    evaluators = [
        {"steps": AB_PLUS_C, "steps_inf": AB, "start": 1, "end": 3, "eq_ind": None},
        {"steps": ABC_PLUS_A, "steps_inf": ABC, "start": 1, "end": 3, "eq_ind": None},
        {"steps": SQ_PLUS, "steps_inf": SQ_INF, "start": 2, "end": 3, "eq_ind": None},
        {"steps": AB_PLUS_C, "steps_inf": AB, "start": 1, "end": 3, "eq_ind": eq},
        {"steps": ab_plus_k, "steps_inf": AB, "start": 1, "end": 2, "eq_ind": eq},
        {"steps": ab_plus_k, "steps_inf": AB, "start": 1, "end": 3, "eq_ind": None},
        {"steps": lin, "steps_inf": lin, "start": 1, "end": 3, "eq_ind": eq},
        {"steps": kab, "steps_inf": kab, "start": 1, "end": 3, "eq_ind": None},
        {"steps": lin, "steps_inf": lin, "start": 1, "end": 3, "eq_ind": None},  # (a lone column at both points: streaming sums from 2^18 points)
    ]
    got, want = both(hal, oracle, 1, n_vars, None, [("folded", v, 0) for v in x], evaluators, [])
    assert got == want


def test_round_evals_routed_with_transparent_inputs(hal, oracle):
    """The routed path on top of materialised Transparent multilinears (the all-ones table shares their scratch block)."""
    n_vars, q_vars, level = 10, 2, 5
    n_ml = n_vars + q_vars
    packed = oracle.random_b128(0x19A0, (1 << n_ml) >> (7 - level))
    query = oracle.arr(1 << q_vars)
    query[0, 0] = 1
    oracle.tensor_expand(query, 0, oracle.random_scalars(0x19B0, q_vars))
    other = [oracle.random_b128(0x19C0 + j, 1 << n_vars) for j in range(2)]
    evaluators = [{"steps": AB_PLUS_C, "steps_inf": AB, "start": 1, "end": 3, "eq_ind": None}]
    mls = [("transparent", packed, level, n_ml), ("folded", other[0], 0), ("folded", other[1], 0)]
    got, want = both(hal, oracle, 1, n_vars, query, mls, evaluators, [])
    assert got == want


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("level", [0, 3, 4, 5, 6, 7])
def test_round_evals_transparent(hal, oracle, order, level):
    n_vars, q_vars = 9, 3
    n_ml = n_vars + q_vars
    packed = oracle.random_b128(0x13A0 + level, (1 << n_ml) >> (7 - level))
    coords = oracle.random_scalars(0x13B0, q_vars)
    query = oracle.arr(1 << q_vars)
    query[0, 0] = 1
    oracle.tensor_expand(query, 0, coords)
    other = oracle.random_b128(0x13C0, 1 << n_vars)
    third = oracle.random_b128(0x13D0, 100)
    evaluators = [{"steps": AB_PLUS_C, "steps_inf": AB, "start": 0, "end": 5, "eq_ind": None}]
    pts = oracle.random_scalars(0x13E0, 2)
    mls = [("transparent", packed, level, n_ml), ("folded", other, 0), ("folded", third, 0x77)]
    got, want = both(hal, oracle, order, n_vars, query, mls, evaluators, pts)
    assert got == want


def test_round_evals_transparent_without_query(hal, oracle):
    """Round 0 of a prover whose multilinears are still transparent: query of zero variables."""
    n_vars = 8
    a, b = oracle.random_b128(0x14A0, 1 << n_vars), oracle.random_b128(0x14B0, (1 << n_vars) >> 2)
    evaluators = [{"steps": AB, "steps_inf": AB, "start": 1, "end": 3, "eq_ind": None}]
    mls = [("transparent", a, 7, n_vars), ("transparent", b, 5, n_vars)]
    got, want = both(hal, oracle, 1, n_vars, None, mls, evaluators, [])
    assert got == want


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("n_vars,length", [(1, 2), (1, 1), (6, 0), (6, 1), (6, 31), (6, 32), (6, 33), (6, 64), (12, 4096), (12, 2049), (17, 1 << 17), (17, 70001)])
def test_fold_folded(hal, oracle, order, n_vars, length):
    alloc = hal.dev_alloc()
    x = oracle.random_b128(0x15A0 + n_vars, 1 << n_vars)[:length].copy()
    sfx, z = oracle.random_scalars(0x15B0, 2)
    d = upload(hal, alloc, x)
    out = alloc.alloc(1 << n_vars)
    n = hal.hal_fold_multilinear(order, n_vars, ("folded", d, sfx), z, None, out)
    rc, want = oracle.hal_fold_multilinear(order, n_vars, ("folded", x, sfx), z)
    assert rc == 0 and n == want.shape[0]
    if n:
        assert np.array_equal(hal.copy_d2h(out.slice(0, n)), want)
    if order == 1 and length == 1 << n_vars:
        # in place, the ComputeLayer shape
        n2 = hal.hal_fold_multilinear(1, n_vars, ("folded", d, sfx), z, None, d)
        assert n2 == n and np.array_equal(hal.copy_d2h(d.slice(0, n)), want)


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("level", [0, 3, 5, 7])
def test_fold_transparent_switchover(hal, oracle, order, level):
    n_vars, q_vars = 10, 4  # the multilinear has n_vars - 1 + q_vars variables
    n_ml = n_vars - 1 + q_vars
    alloc = hal.dev_alloc()
    packed = oracle.random_b128(0x16A0 + level, (1 << n_ml) >> (7 - level))
    query = oracle.arr(1 << q_vars)
    query[0, 0] = 1
    oracle.tensor_expand(query, 0, oracle.random_scalars(0x16B0, q_vars))
    out = alloc.alloc(1 << (n_vars - 1))
    n = hal.hal_fold_multilinear(order, n_vars, ("transparent", upload(hal, alloc, packed), level, n_ml), 0, upload(hal, alloc, query), out)
    rc, want = oracle.hal_fold_multilinear(order, n_vars, ("transparent", packed, level, n_ml), 0, query)
    assert rc == 0 and n == want.shape[0] == 1 << (n_vars - 1)
    assert np.array_equal(hal.copy_d2h(out), want)


def test_sumcheck_through_the_old_hal_matches_v3(hal, oracle):
    """A whole High-to-Low bivariate-product sumcheck driven through the two old-HAL calls gives the v3 prover's
    transcript (the two interfaces front the same mathematics: crates/core/src/protocols/sumcheck/prove/
    regular_sumcheck.rs vs v3/bivariate_product.rs)."""
    from binius_amd._ffi import HostField as F

    n_vars = 11
    alloc = hal.dev_alloc()
    x = [oracle.random_b128(0x17A0 + j, 1 << n_vars) for j in range(2)]
    d = [upload(hal, alloc, v) for v in x]
    rc, claim = oracle.inner_product(x[0], 7, x[1])
    stream = oracle.random_scalars(0x17B0, n_vars + 1)
    want_coeffs, want_final = oracle.bivariate_sumcheck_prove([v.copy() for v in x], n_vars, [(0, 1)], [claim], stream[0], stream[1:])
    c = hal.compile_expr(AB)
    sum_ = claim
    lens = [1 << n_vars] * 2
    for r in range(n_vars):
        nv = n_vars - r
        (y1, yinf), = hal.hal_round_evals(1, nv, None, [("folded", d[j].slice(0, lens[j]), 0) for j in range(2)],
                                          [{"composition": c, "composition_at_infinity": c, "start": 1, "end": 3, "eq_ind": None}], [])
        y0 = sum_ ^ y1
        coeffs = [y0, y1 ^ y0 ^ yinf, yinf]
        assert coeffs == list(want_coeffs[r])
        z = stream[1 + r]
        sum_ = coeffs[0] ^ F.mul(z, coeffs[1]) ^ F.mul(F.mul(z, z), coeffs[2])
        for j in range(2):
            lens[j] = hal.hal_fold_multilinear(1, nv, ("folded", d[j].slice(0, lens[j]), 0), z, None, d[j])
    c.free()
    assert [int.from_bytes(hal.copy_d2h(d[j].slice(0, 1)).tobytes(), "little") for j in range(2)] == list(want_final)


def test_error_behaviour(hal, oracle):
    from binius_amd import BnError

    alloc = hal.dev_alloc()
    x = upload(hal, alloc, oracle.random_b128(1, 16))
    c = hal.compile_expr(AB)
    ev = {"composition": c, "composition_at_infinity": c, "start": 0, "end": 5, "eq_ind": None}
    mls = [("folded", x, 0), ("folded", x, 0)]
    with pytest.raises(BnError) as e:  # Error::IncorrectNontrivialEvalPointsLength
        hal.hal_round_evals(1, 4, None, mls, [ev], [7])
    assert e.value.kind == "InputValidation"
    with pytest.raises(BnError):  # no variables left
        hal.hal_round_evals(1, 0, None, mls, [ev], [7, 9])
    with pytest.raises(BnError):  # composition over more variables than multilinears
        hal.hal_round_evals(1, 4, None, mls[:1], [ev], [7, 9])
    with pytest.raises(BnError):  # transparent multilinear whose size does not match n_vars + query variables
        hal.hal_round_evals(1, 4, None, [("transparent", x, 7, 5), ("folded", x, 0)], [ev], [7, 9])
    out = alloc.alloc(16)
    with pytest.raises(BnError):  # output too small
        hal.hal_fold_multilinear(1, 4, ("folded", x, 0), 3, None, out.slice(0, 4))
    with pytest.raises(BnError):  # Low-to-High in place
        hal.hal_fold_multilinear(0, 4, ("folded", x, 0), 3, None, x)
    # a failed call leaves the context usable
    assert hal.hal_round_evals(1, 4, None, mls, [dict(ev, end=3)], []) == oracle.hal_round_evals(
        1, 4, None, [("folded", oracle.random_b128(1, 16), 0)] * 2, [{"steps": AB, "steps_inf": AB, "start": 0, "end": 3, "eq_ind": None}], [])[1]
    c.free()
