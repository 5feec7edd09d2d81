"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's eq-ind sumcheck prover over its old hardware abstraction layer,
composed from the oracle's pinned pieces (oracle/hal_ref.c through oracle/__init__.py: CpuBackend's round calculation and fold,
pinned by tests/test_oracle_hal.py; the scalar field, pinned by the reference's known answers).  Only tests/ import this; the product path and tools/ never do.

  eqind_sumcheck_prove   EqIndSumcheckProver::{execute, fold, finish}   crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644
                         ProverState::{calculate_round_evals, fold}      crates/core/src/protocols/sumcheck/prove/prover_state.rs:138-265
                         fold_partial_eq_ind                              crates/core/src/protocols/sumcheck/prove/common.rs:13-75
                         Interpolator::round_evals_to_coeffs              eq_ind.rs:753-779
                         InterpolationDomain::interpolate                 crates/math/src/univariate.rs:60-99, 227-236, 281-300

Evaluation order High-to-Low, compositions of degree d >= 1: evaluation points 1 ..= d (eq_ind.rs:664-668) = X = 1, infinity
(d >= 2), and the points 2 .. d - 1 of the default interpolation domain -- the first d elements 0, 1, 2, ... of the binary subspace
plus infinity; the prime polynomial is the solution of the Vandermonde system with the infinity row.

Parity pin: the pieces are pinned as said; the bookkeeping added here is checked in tests/test_gpu_zerocheck.py against the
VERIFIER's equations (protocols/sumcheck/verify.rs, eq_ind.rs verify side): every round polynomial sums to the running claim over
{0, 1}, and the last claim is the batched composition of the final evaluations times the indicator's evaluation."""
import numpy as np

import oracle as o


def interpolate(finite, at_infinity):
    """Coefficients c_0 .. c_d of the polynomial of degree <= d = len(finite) with P(i) = finite[i] at the domain's finite points
    i = 0 .. d - 1 (tower elements of those integer values) and leading coefficient at_infinity (d >= 2): the Vandermonde system of
    univariate.rs:281-300 solved by elimination -- plain linear algebra, no shortcut shared with the code under test."""
    d = len(finite)
    n = d + 1
    rows = []
    for i in range(d):
        row, acc = [], 1
        for _ in range(n):
            row.append(acc)
            acc = o.mul(acc, i)
        rows.append(row + [finite[i]])
    rows.append([0] * (n - 1) + [1, at_infinity])
    for col in range(n):  # Gauss-Jordan over the field
        piv = next(r for r in range(col, n) if rows[r][col])
        rows[col], rows[piv] = rows[piv], rows[col]
        inv = o.invert(rows[col][col])
        rows[col] = [o.mul(v, inv) for v in rows[col]]
        for r in range(n):
            if r != col and rows[r][col]:
                f = rows[r][col]
                rows[r] = [a ^ o.mul(f, b) for a, b in zip(rows[r], rows[col])]
    return [rows[i][n] for i in range(n)]


def eqind_sumcheck_prove(multilins, n_vars, compositions, sums, eq_ind_challenges, batch_coeff, challenges, degrees=None):
    """multilins: numpy arrays of 2^n_vars elements (copied); compositions: [(steps, steps_of_the_leading_form)]; degrees: per
    composition (None: all 2) -- the evaluation points are 1 ..= degree (eq_ind.rs:664-668), a linear composition's prime
    polynomial is interpolated from R'(0), R'(1) alone (:753-779).
    Returns (round_coeffs[n_vars][D + 2], final_evals[m + 1]), D = max(2, largest degree)."""
    assert len(eq_ind_challenges) == n_vars
    mls = [x.copy() for x in multilins]
    # eq_ind_expand (eq_ind.rs:430-446): the tensor expansion of all challenges but the last
    eq = o.arr(1 << (n_vars - 1))
    eq[0] = o.ints_to_arr([1])[0]
    o.tensor_expand(eq, 0, list(eq_ind_challenges[: n_vars - 1]))
    sums = list(sums)
    degrees = list(degrees) if degrees is not None else [2] * len(compositions)
    D = max([2] + degrees)
    points = list(range(2, D))  # the nontrivial evaluation points (sumcheck/common.rs:310-340): finite points 2 .. D - 1
    prefix = 1
    out = []
    for r in range(n_vars):
        n_rem = n_vars - r
        alpha = eq_ind_challenges[n_vars - 1 - r]
        evaluators = [{"steps": c, "steps_inf": ci, "start": 1, "end": 1 + d, "eq_ind": eq[: 1 << (n_rem - 1)]} for (c, ci), d in zip(compositions, degrees)]
        rc, evals = o.hal_round_evals(1, n_rem, None, [("folded", np.ascontiguousarray(x[: 1 << n_rem]), 0) for x in mls], evaluators, points)
        assert rc == 0
        denom_inv = o.invert(1 ^ alpha) if (1 ^ alpha) else 0
        prime, batched, scale = [], [0] * (D + 1), 1
        for c in range(len(compositions)):
            d = degrees[c]
            y1, yinf = evals[c][0], (evals[c][1] if d >= 2 else 0)
            y0 = o.mul(sums[c] ^ o.mul(y1, alpha), denom_inv)
            if d <= 2:
                pc = [y0, y1 ^ y0 ^ yinf, yinf]
            else:
                pc = interpolate([y0, y1] + list(evals[c][2:d]), yinf)
            pc = pc + [0] * (D + 1 - len(pc))
            prime.append(pc)
            for i in range(D + 1):
                batched[i] ^= o.mul(pc[i], scale)
            scale = o.mul(scale, batch_coeff)
        coeffs = [0] * (D + 2)
        for i in range(D + 1):
            coeffs[i] ^= o.mul(batched[i], 1 ^ alpha)
            coeffs[i + 1] ^= batched[i]
        out.append([o.mul(v, prefix) for v in coeffs])
        z = challenges[r]
        prefix = o.mul(prefix, alpha ^ z ^ 1)  # eq(alpha, z) in characteristic 2 (field/src/util.rs:72-81)
        sums = [o.evaluate_univariate(pc, z) for pc in prime]
        half = 1 << (n_rem - 1)
        for x in mls:
            lo, hi = np.ascontiguousarray(x[:half]), np.ascontiguousarray(x[half : 2 * half])
            o.extrapolate_line(lo, hi, z)
            x[:half] = lo
        if n_rem - 1 > 0:
            q = half >> 1
            eq[:q] ^= eq[q:half]
    finals = [o.arr_to_ints(x[:1])[0] for x in mls] + [prefix]
    return out, finals
