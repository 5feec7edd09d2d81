"""Pins the oracle's restatement of the eq-ind sumcheck prover over the old HAL (oracle/zerocheck_ref.py; reference:
crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644) by the VERIFIER's equations: every round polynomial sums to the
running claim over {0, 1}; the last claim is the batched composition of the final evaluations times the indicator's
evaluation; the final evaluations are the multilinear extensions at the reversed challenges (High-to-Low binds the top variable
first).  CPU only."""
import pytest


DEGREE_2 = [
    ([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3)], [("var", 0), ("var", 1), ("mul", 0, 1)]),
    ([("var", 3), ("var", 4), ("add", 0, 1), ("var", 5), ("mul", 2, 3)], [("var", 3), ("var", 4), ("add", 0, 1), ("var", 5), ("mul", 2, 3)]),
    ([("var", 1), ("const", 1), ("add", 0, 1), ("var", 2), ("mul", 2, 3), ("var", 0), ("add", 4, 5), ("var", 4), ("add", 6, 7)], [("var", 1), ("var", 2), ("mul", 0, 1)]),
]
# (the u32_add table's zout constraint, m3/src/gadgets/add.rs:104-110, between two of degree 2; a linear one with a constant factor)
LINEAR = [("var", 0), ("var", 1), ("add", 0, 1), ("var", 2), ("add", 2, 3), ("var", 4), ("add", 4, 5)]
SCALED = [("var", 3), ("const", 0x1234567890ABCDEF1122334455667788), ("mul", 0, 1), ("var", 5), ("add", 2, 3)]
MIXED = [DEGREE_2[0], (LINEAR, LINEAR), DEGREE_2[1], (SCALED, SCALED)]
# degree 3 and 4 (evaluation points 1, infinity, 2 and 3 of the interpolation domain: eq_ind.rs:664-668, math/src/univariate.rs:60-99)
ABC = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)]
ABC_PLUS = ABC + [("var", 3), ("var", 4), ("mul", 5, 6), ("add", 4, 7), ("var", 5), ("add", 8, 9)]
ABCD = [("var", 2), ("var", 3), ("mul", 0, 1), ("var", 4), ("mul", 2, 3), ("var", 5), ("mul", 4, 5)]
ABCD_PLUS = ABCD + [("var", 0), ("add", 6, 7)]
CUBIC = [(ABC_PLUS, ABC), DEGREE_2[0], (LINEAR, LINEAR), (ABC, ABC)]
QUARTIC = [(ABCD_PLUS, ABCD), (ABC_PLUS, ABC), DEGREE_2[1]]


@pytest.mark.parametrize("n_vars", [1, 2, 5, 7])
@pytest.mark.parametrize("comps,degrees", [(DEGREE_2, None), (MIXED, [2, 1, 2, 1]), (CUBIC, [3, 2, 1, 3]), (QUARTIC, [4, 3, 2])], ids=["degree2", "mixed", "cubic", "quartic"])
def test_restatement_satisfies_the_verifier(oracle, n_vars, comps, degrees):
    from oracle import zerocheck_ref as z

    o = oracle
    m = 6
    mls = [o.random_b128(0x2C00 + 16 * n_vars + j, 1 << n_vars) for j in range(m)]
    eqc, ch, bc = o.random_scalars(0x2C10 + n_vars, n_vars), o.random_scalars(0x2C20 + n_vars, n_vars), o.random_scalars(0x2C30, 1)[0]
    eq_full = o.arr(1 << n_vars)
    eq_full[0] = o.ints_to_arr([1])[0]
    o.tensor_expand(eq_full, 0, list(eqc))
    eqi = o.arr_to_ints(eq_full)
    vals = [o.arr_to_ints(x) for x in mls]
    sums = []
    for c, _ in comps:
        s = 0
        for i in range(1 << n_vars):
            s ^= o.mul(eqi[i], o.circuit_eval(c, [v[i] for v in vals]))
        sums.append(s)
    coeffs, fin = z.eqind_sumcheck_prove(mls, n_vars, comps, sums, eqc, bc, ch, degrees)
    running = o.evaluate_univariate(sums, bc)
    for r in range(n_vars):
        c = coeffs[r]
        p1 = 0
        for v in c:
            p1 ^= v
        assert len(c) == 2 + max([2] + (degrees or [])) and c[0] ^ p1 == running, "round %d: P(0) + P(1) is not the running claim" % r
        running = o.evaluate_univariate(c, ch[r])
    acc, p = 0, 1
    for c, _ in comps:
        acc ^= o.mul(p, o.circuit_eval(c, fin[:m]))
        p = o.mul(p, bc)
    assert o.mul(acc, fin[m]) == running
    point = list(reversed(ch))
    for j in range(m):
        assert o.mle_evaluate(mls[j], n_vars, point) == fin[j]
    # the prefix is the indicator at (challenges reversed) against the indicator's point
    assert o.mle_evaluate(eq_full, n_vars, point) == fin[m]
