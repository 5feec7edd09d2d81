"""Pins of the oracle's restatement of the OLD HAL (oracle/hal_ref.c: binius_hal::ComputationBackend,
crates/hal/src/sumcheck_round_calculation.rs:45-330, crates/hal/src/sumcheck_folding.rs:16-237).

The reference holds no known-answer vectors for this path (its tests are prove -> verify round trips), so the
restatement is pinned by what those round trips check plus independent recomputation:
  * a pure-Python evaluation of the definition (one hypercube point at a time, scalar oracle.mul pinned by the
    tower golden vectors) for every order / multilinear kind / evaluation point;
  * the verifier's equations: R(0) + R(1) = claimed sum, and the polynomial interpolated from R(0), R(1),
    R(infinity) reproduces R at a further domain point (crates/core/src/protocols/sumcheck/prove/
    regular_sumcheck.rs:188-239: infinity = leading coefficient);
  * the already-pinned v3 functions: High-to-Low products = oracle.round_evals, fold = extrapolate_line,
    Transparent = Folded(fold_left / fold_right of the packed values)."""
import numpy as np
import pytest

import oracle

L2H, H2L = oracle.ORDER_LOW_TO_HIGH, oracle.ORDER_HIGH_TO_LOW
AB = [("var", 0), ("var", 1), ("mul", 0, 1)]
AB_PLUS_C = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3)]
ABC_PLUS_A = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3), ("add", 4, 0)]
ABC = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)]


@pytest.fixture(scope="module", autouse=True)
def _built():
    oracle.build()


def _get(x, length, suffix, i):
    return x[i] if i < length else suffix


def py_round_evals(order, n_vars, cols, evaluators, points):
    """cols: per multilinear (list of ints, stored length, suffix).  The definition, one point at a time."""
    half = 1 << (n_vars - 1)
    res = []
    for e in evaluators:
        vals = []
        for p in range(e["start"], e["end"]):
            acc = 0
            for i in range(half):
                i0, i1 = (2 * i, 2 * i + 1) if order == L2H else (i, i + half)
                row = []
                for x, ln, sfx in cols:
                    e0, e1 = _get(x, ln, sfx, i0), _get(x, ln, sfx, i1)
                    if p == 0:
                        row.append(e0)
                    elif p == 1:
                        row.append(e1)
                    elif p == 2:
                        row.append(e0 ^ e1)
                    else:
                        row.append(e0 ^ oracle.mul(points[p - 3], e0 ^ e1))
                v = oracle.circuit_eval(e["steps_inf"] if p == 2 else e["steps"], row)
                if e.get("eq_ind") is not None:
                    v = oracle.mul(v, oracle.arr_to_ints(e["eq_ind"])[i])
                acc ^= v
            vals.append(acc)
        res.append(vals)
    return res


@pytest.mark.parametrize("order", [L2H, H2L])
@pytest.mark.parametrize("n_vars", [1, 2, 4, 5])
def test_round_evals_equal_the_definition(order, n_vars):
    n = 1 << n_vars
    mls = [oracle.random_b128(0x5100 + j, n) for j in range(3)]
    lens = [n, max(1, n - 3), n // 2]
    sfx = oracle.random_scalars(0x5200, 3)
    sfx[0] = 0
    eq = oracle.random_b128(0x5300, max(1, n // 2))
    pts = oracle.random_scalars(0x5400, 2)
    evaluators = [
        {"steps": AB_PLUS_C, "steps_inf": AB, "start": 0, "end": 4, "eq_ind": None},
        {"steps": ABC_PLUS_A, "steps_inf": ABC, "start": 1, "end": 5, "eq_ind": eq},
        {"steps": AB, "steps_inf": AB, "start": 2, "end": 3, "eq_ind": None},
    ]
    folded = [("folded", np.ascontiguousarray(mls[j][: lens[j]]), sfx[j]) for j in range(3)]
    rc, got = oracle.hal_round_evals(order, n_vars, None, folded, evaluators, pts)
    assert rc == 0
    cols = [(oracle.arr_to_ints(mls[j]), lens[j], sfx[j]) for j in range(3)]
    assert got == py_round_evals(order, n_vars, cols, evaluators, pts)


@pytest.mark.parametrize("order", [L2H, H2L])
def test_verifier_equations(order):
    n_vars = 6
    n = 1 << n_vars
    mls = [oracle.random_b128(0x6100 + j, n) for j in range(3)]
    z = oracle.random_scalars(0x6200, 1)
    ev = [{"steps": AB_PLUS_C, "steps_inf": AB, "start": 0, "end": 4, "eq_ind": None}]
    rc, got = oracle.hal_round_evals(order, n_vars, None, [("folded", x, 0) for x in mls], ev, z)
    assert rc == 0
    r0, r1, rinf, rz = got[0]
    # claimed sum of a*b + c over the cube
    want = 0
    A, B, Cc = [oracle.arr_to_ints(x) for x in mls]
    for i in range(n):
        want ^= oracle.mul(A[i], B[i]) ^ Cc[i]
    assert r0 ^ r1 == want
    # R(X) = c0 + c1 X + c2 X^2 with c2 = R(infinity)
    c0, c2 = r0, rinf
    c1 = r1 ^ c0 ^ c2
    assert rz == c0 ^ oracle.mul(c1, z[0]) ^ oracle.mul(c2, oracle.mul(z[0], z[0]))


def test_high_to_low_products_equal_v3_round_evals():
    n_vars = 9
    mls = [oracle.random_b128(0x7100 + j, 1 << n_vars) for j in range(3)]
    comps = [(0, 1), (2, 0)]
    evs = [{"steps": [("var", a), ("var", b), ("mul", 0, 1)], "steps_inf": [("var", a), ("var", b), ("mul", 0, 1)], "start": 1, "end": 3, "eq_ind": None}
           for a, b in comps]
    rc, got = oracle.hal_round_evals(H2L, n_vars, None, [("folded", x, 0) for x in mls], evs, [])
    assert rc == 0
    # per-composition sums through the pinned v3 function: batch coefficient 1 on a single composition at a time
    for k, (a, b) in enumerate(comps):
        rc, want = oracle.round_evals([mls[a], mls[b]], n_vars, [(0, 1)], 1)
        assert rc == 0 and got[k] == want


@pytest.mark.parametrize("order", [L2H, H2L])
def test_low_to_high_is_high_to_low_of_the_deinterleaved_array(order):
    n_vars = 7
    n = 1 << n_vars
    mls = [oracle.random_b128(0x8100 + j, n) for j in range(2)]
    ev = [{"steps": AB, "steps_inf": AB, "start": 0, "end": 3, "eq_ind": None}]
    other = H2L if order == L2H else L2H
    if order == L2H:
        perm = [np.ascontiguousarray(np.concatenate([x[0::2], x[1::2]])) for x in mls]
    else:
        perm = []
        for x in mls:
            y = np.empty_like(x)
            y[0::2], y[1::2] = x[: n // 2], x[n // 2 :]
            perm.append(y)
    rc0, a = oracle.hal_round_evals(order, n_vars, None, [("folded", x, 0) for x in mls], ev, [])
    rc1, b = oracle.hal_round_evals(other, n_vars, None, [("folded", x, 0) for x in perm], ev, [])
    assert rc0 == 0 and rc1 == 0 and a == b


@pytest.mark.parametrize("order", [L2H, H2L])
@pytest.mark.parametrize("level", [0, 3, 5, 7])
def test_transparent_equals_folded_partial_evaluation(order, level):
    n_vars, q_vars = 5, 3
    n_ml = n_vars + q_vars
    words = max(1, (1 << n_ml) >> (7 - level))
    packed = oracle.random_b128(0x9100 + level, words)
    coords = oracle.random_scalars(0x9200, q_vars)
    query = oracle.arr(1 << q_vars)
    # query expansion through the pinned tensor_expand: start from [1]
    query[0, 0] = 1
    oracle.tensor_expand(query, 0, coords)
    full = oracle.arr(1 << n_vars)
    (oracle.fold_right if order == L2H else oracle.fold_left)(packed, level, query, full)
    other = oracle.random_b128(0x9300, 1 << n_vars)
    ev = [{"steps": AB, "steps_inf": AB, "start": 0, "end": 4, "eq_ind": None}]
    pts = oracle.random_scalars(0x9400, 1)
    rc0, a = oracle.hal_round_evals(order, n_vars, query, [("transparent", packed, level, n_ml), ("folded", other, 0)], ev, pts)
    rc1, b = oracle.hal_round_evals(order, n_vars, None, [("folded", full, 0), ("folded", other, 0)], ev, pts)
    assert rc0 == 0 and rc1 == 0 and a == b
    # switchover fold: the query already holds the round challenge -> one variable fewer
    full_next = oracle.arr(1 << (n_vars + 1))
    (oracle.fold_right if order == L2H else oracle.fold_left)(packed, level, query[: 1 << (q_vars - 1)].copy(), full_next)
    rc, got = oracle.hal_fold_multilinear(order, n_vars + 2, ("transparent", packed, level, n_ml), 0, query[: 1 << (q_vars - 1)].copy())
    assert rc == 0 and np.array_equal(got, full_next)


@pytest.mark.parametrize("order", [L2H, H2L])
@pytest.mark.parametrize("length", [0, 1, 5, 31, 32, 33, 64])
def test_fold_with_suffix_equals_fold_of_the_padded_array(order, length):
    n_vars = 6
    n = 1 << n_vars
    x = oracle.random_b128(0xA100, n)
    sfx, z = oracle.random_scalars(0xA200, 2)
    padded = x.copy()
    padded[length:] = oracle.ints_to_arr([sfx])[0]
    rc, got = oracle.hal_fold_multilinear(order, n_vars, ("folded", np.ascontiguousarray(x[:length]) if length else oracle.arr(1)[:0], sfx), z)
    assert rc == 0
    P = oracle.arr_to_ints(padded)
    half = n // 2
    want = []
    for i in range(half):
        e0, e1 = (P[2 * i], P[2 * i + 1]) if order == L2H else (P[i], P[i + half])
        want.append(e0 ^ oracle.mul(z, e0 ^ e1))
    n_out = (length + 1) // 2 if order == L2H else min(length, half)
    assert got.shape[0] == n_out
    assert oracle.arr_to_ints(got) == want[:n_out]
    # beyond the stored prefix the folded multilinear is again the constant suffix (lerp of two equal values)
    if order == L2H or length <= half:
        assert all(w == sfx for w in want[n_out:])


def test_high_to_low_full_fold_is_extrapolate_line():
    n_vars = 8
    x = oracle.random_b128(0xB100, 1 << n_vars)
    z = oracle.random_scalars(0xB200, 1)[0]
    rc, got = oracle.hal_fold_multilinear(H2L, n_vars, ("folded", x, 0), z)
    assert rc == 0
    lo, hi = x[: 1 << (n_vars - 1)].copy(), x[1 << (n_vars - 1) :].copy()
    assert oracle.extrapolate_line(lo, hi, z) == 0
    assert np.array_equal(got, lo)


def test_error_behaviour():
    x = oracle.random_b128(1, 4)
    ev = [{"steps": AB, "steps_inf": AB, "start": 0, "end": 5, "eq_ind": None}]
    # IncorrectNontrivialEvalPointsLength (sumcheck_round_calculation.rs:121-125): 5 points need 2 nontrivial ones
    rc, _ = oracle.hal_round_evals(H2L, 2, None, [("folded", x, 0), ("folded", x, 0)], ev, [7])
    assert rc != 0
    # zero variables: nothing to evaluate a round over
    rc, _ = oracle.hal_round_evals(H2L, 0, None, [("folded", x, 0), ("folded", x, 0)], ev, [7, 9])
    assert rc != 0
