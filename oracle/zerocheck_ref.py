"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's eq-ind sumcheck prover over its old hardware abstraction layer,
composed from the oracle's pinned pieces (oracle/hal_ref.c through oracle/__init__.py: CpuBackend's round calculation and fold,
pinned by tests/test_oracle_hal.py; the scalar field, pinned by the reference's known answers).  Only tests/ import this; the product path and tools/ never do.

  eqind_sumcheck_prove   EqIndSumcheckProver::{execute, fold, finish}   crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644
                         ProverState::{calculate_round_evals, fold}      crates/core/src/protocols/sumcheck/prove/prover_state.rs:138-265
                         fold_partial_eq_ind                              crates/core/src/protocols/sumcheck/prove/common.rs:13-75
                         Interpolator::round_evals_to_coeffs (degree 1, 2) eq_ind.rs:753-779

Evaluation order High-to-Low, compositions of degree 1 or 2 (evaluation points 1 ..= degree: 1 and, for degree 2, infinity;
eq_ind.rs:664-668).

Parity pin: the pieces are pinned as said; the bookkeeping added here is checked in tests/test_gpu_zerocheck.py against the
VERIFIER's equations (protocols/sumcheck/verify.rs, eq_ind.rs verify side): every round polynomial sums to the running claim over
{0, 1}, and the last claim is the batched composition of the final evaluations times the indicator's evaluation."""
import numpy as np

import oracle as o


def eqind_sumcheck_prove(multilins, n_vars, compositions, sums, eq_ind_challenges, batch_coeff, challenges, degrees=None):
    """multilins: numpy arrays of 2^n_vars elements (copied); compositions: [(steps, steps_of_the_leading_form)]; degrees: 1 or 2 per
    composition (None: all 2) -- the evaluation points are 1 ..= degree (eq_ind.rs:664-668), a linear composition's prime
    polynomial is interpolated from R'(0), R'(1) alone (:753-779).
    Returns (round_coeffs[n_vars][4], final_evals[m + 1])."""
    assert len(eq_ind_challenges) == n_vars
    mls = [x.copy() for x in multilins]
    # eq_ind_expand (eq_ind.rs:430-446): the tensor expansion of all challenges but the last
    eq = o.arr(1 << (n_vars - 1))
    eq[0] = o.ints_to_arr([1])[0]
    o.tensor_expand(eq, 0, list(eq_ind_challenges[: n_vars - 1]))
    sums = list(sums)
    degrees = list(degrees) if degrees is not None else [2] * len(compositions)
    prefix = 1
    out = []
    for r in range(n_vars):
        n_rem = n_vars - r
        alpha = eq_ind_challenges[n_vars - 1 - r]
        evaluators = [{"steps": c, "steps_inf": ci, "start": 1, "end": 1 + d, "eq_ind": eq[: 1 << (n_rem - 1)]} for (c, ci), d in zip(compositions, degrees)]
        rc, evals = o.hal_round_evals(1, n_rem, None, [("folded", np.ascontiguousarray(x[: 1 << n_rem]), 0) for x in mls], evaluators, [])
        assert rc == 0
        denom_inv = o.invert(1 ^ alpha) if (1 ^ alpha) else 0
        prime, batched, scale = [], [0, 0, 0], 1
        for c in range(len(compositions)):
            y1, yinf = evals[c][0], (evals[c][1] if degrees[c] == 2 else 0)
            y0 = o.mul(sums[c] ^ o.mul(y1, alpha), denom_inv)
            pc = [y0, y1 ^ y0 ^ yinf, yinf]
            prime.append(pc)
            for i in range(3):
                batched[i] ^= o.mul(pc[i], scale)
            scale = o.mul(scale, batch_coeff)
        coeffs = [0, 0, 0, 0]
        for i in range(3):
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
