"""Pin the oracle's tower arithmetic against every known-answer vector the reference holds
(tests/golden/field_kats.json, extracted by tests/golden/gen_field_kats.py from
crates/field/src/binary_field.rs:740-747,931-1029 and crates/field/src/polyval.rs:516-784,
1112-1127)."""
import json
import os
import random

import numpy as np
import pytest

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "field_kats.json")))
LEVEL = {1: 0, 2: 1, 4: 2, 8: 3, 16: 4, 32: 5, 64: 6, 128: 7}
M128 = (1 << 128) - 1


def test_tower_mul_kats(oracle):
    n = 0
    for bits, rows in KATS["tower_mul_kats"].items():
        lvl = LEVEL[int(bits)]
        for a, b, c in rows:
            a, b, c = int(a, 16), int(b, 16), int(c, 16)
            assert oracle.gf_mul(a, b, lvl) == c, (bits, hex(a), hex(b))
            assert oracle.gf_mul_slow(a, b, lvl) == c
            n += 1
    assert n == 50


def test_mul8_table_matches_recursion(oracle):
    for a in range(256):
        for b in range(0, 256, 7):
            assert oracle.gf_mul(a, b, 3) == oracle.gf_mul_slow(a, b, 3)


def _pow(oracle, g, e, bits):
    lvl = LEVEL[bits]
    if bits == 128:
        mul = oracle.mul
    else:
        mul = lambda x, y: oracle.gf_mul(x, y, lvl)
    r, base = 1, g
    while e:
        if e & 1:
            r = mul(r, base)
        base = mul(base, base)
        e >>= 1
    return r


FACTORS = {
    1: [],
    2: [3],
    4: [3, 5],
    8: [3, 5, 17],
    16: [3, 5, 17, 257],
    32: [3, 5, 17, 257, 65537],
    64: [3, 5, 17, 257, 641, 65537, 6700417],
    128: [3, 5, 17, 257, 641, 65537, 274177, 6700417, 67280421310721],
}


@pytest.mark.parametrize("bits", [2, 4, 8, 16, 32, 64, 128])
def test_multiplicative_generators(oracle, bits):
    """binary_field.rs:740-747 + is_binary_field_valid_generator :1031-1090: g has full order."""
    g = int(KATS["multiplicative_generators"][str(bits)], 16)
    order = (1 << bits) - 1
    prod = 1
    for p in FACTORS[bits]:
        prod *= p
    assert prod == order
    assert _pow(oracle, g, order, bits) == 1
    for p in FACTORS[bits]:
        assert _pow(oracle, g, order // p, bits) != 1


# ---- POLYVAL side, restated here in Python (test-only) ----
POLY = (1 << 128) | (1 << 127) | (1 << 126) | (1 << 121) | 1  # polyval.rs:262 modulus


def clmul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return r


def polymod(c):
    for i in range(c.bit_length() - 1, 127, -1):
        if (c >> i) & 1:
            c ^= POLY << (i - 128)
    return c


def mont_mul(a, b):
    """a*b*x^-128 mod POLY (polyval.rs Montgomery multiplication)."""
    c = clmul(a, b)
    for i in range(128):
        if (c >> i) & 1:
            c ^= POLY << i
    return c >> 128


def test_polyval_kat_pins_modulus():
    a, b, c = (int(x, 16) for x in KATS["polyval_mul_kat"])
    assert polymod(clmul(a, b)) == c
    a, s = (int(x, 16) for x in KATS["polyval_sqr_kat"])
    assert polymod(clmul(a, a)) == s
    one = int(KATS["polyval_one_montgomery"], 16)
    assert one == polymod(1 << 128)
    r2 = int(KATS["polyval_to_montgomery_factor"], 16)
    assert mont_mul(1, r2) == one  # to_montgomery(1) == ONE


def phi(table, x):
    r = 0
    i = 0
    while x:
        if x & 1:
            r ^= table[i]
        x >>= 1
        i += 1
    return r


def test_b128_mul_is_isomorphic_to_polyval(oracle):
    """polyval.rs test_to_from_tower_basis: from(a_polyval * b_polyval) == a_tower * b_tower.
    This is the reference's only hard constraint on 128-bit tower products."""
    fwd = [int(x, 16) for x in KATS["binary_to_polyval"]]
    inv = [int(x, 16) for x in KATS["polyval_to_binary"]]
    rng = random.Random(1234)
    cases = [(1, 1), (2, 3), (1 << 127, 1 << 127), (M128, M128), (1 << 64, 1 << 64)]
    cases += [(rng.getrandbits(128), rng.getrandbits(128)) for _ in range(300)]
    for a, b in cases:
        assert phi(inv, phi(fwd, a)) == a
        t = oracle.mul(a, b)
        assert phi(fwd, t) == mont_mul(phi(fwd, a), phi(fwd, b))
        assert phi(inv, mont_mul(phi(fwd, a), phi(fwd, b))) == t
    for i in range(128):
        for j in (0, 1, 63, 64, 127):
            a, b = 1 << i, 1 << j
            assert phi(fwd, oracle.mul(a, b)) == mont_mul(fwd[i], fwd[j])


def test_field_axioms_128(oracle):
    rng = random.Random(7)
    for _ in range(100):
        a, b, c = (rng.getrandbits(128) for _ in range(3))
        assert oracle.mul(a, b) == oracle.mul(b, a)
        assert oracle.mul(oracle.mul(a, b), c) == oracle.mul(a, oracle.mul(b, c))
        assert oracle.mul(a, b ^ c) == oracle.mul(a, b) ^ oracle.mul(a, c)
        assert oracle.square(a) == oracle.mul(a, a)
        if a:
            assert oracle.mul(a, oracle.invert(a)) == 1
    assert oracle.invert(0) == 0


def test_subfield_mul(oracle):
    """binary_field.rs:361-412: subfield scalar multiplies limb-wise == full mul by the embedded element."""
    rng = random.Random(9)
    for iota in (0, 3, 4, 5, 6, 7):
        for _ in range(50):
            a = rng.getrandbits(128)
            s = rng.getrandbits(1 << iota)
            assert oracle.mul_subfield(a, s, iota) == oracle.mul(a, s)


def test_small_levels_consistent(oracle):
    rng = random.Random(11)
    for lvl in range(1, 7):
        bits = 1 << lvl
        for _ in range(100):
            a, b = rng.getrandbits(bits), rng.getrandbits(bits)
            p = oracle.gf_mul(a, b, lvl)
            assert p == oracle.gf_mul_slow(a, b, lvl)
            assert oracle.mul(a, b) == p  # subfield embedding is the identity on low bits
            assert oracle.gf_square(a, lvl) == oracle.gf_mul(a, a, lvl)
            if a:
                assert oracle.gf_mul(a, oracle.gf_invert(a, lvl), lvl) == 1
            # mul_alpha_k == multiplication by X_{k-1} = 1 << 2^(k-1)
            assert oracle.gf_mul_alpha(a, lvl) == oracle.gf_mul(a, 1 << (bits // 2), lvl)


def _mlecheck_instance(oracle, n_vars, m, comps, seed):
    mls = [oracle.random_b128(seed + j, 1 << n_vars) for j in range(m)]
    eq_ch = oracle.random_scalars(seed ^ 0xE9, n_vars)
    # eq indicator over all n variables (claims) and over the first n-1 (the prover's table)
    full = oracle.arr(1 << n_vars)
    full[0] = (1, 0)
    oracle.tensor_expand(full, 0, eq_ch)
    sums = []
    for i, j in comps:
        p = oracle.mul_vec(oracle.mul_vec(mls[i], mls[j]), full)
        sums.append(int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64))
    eq = oracle.arr(1 << max(n_vars - 1, 0))
    eq[0] = (1, 0)
    oracle.tensor_expand(eq, 0, eq_ch[: max(n_vars - 1, 0)])
    return mls, eq_ch, eq, sums


@pytest.mark.parametrize("n_vars,m,comps", [(1, 2, [(0, 1)]), (4, 2, [(0, 1)]), (8, 4, [(0, 1), (2, 3), (1, 1)])])
def test_oracle_mlecheck_prover_verifies(oracle, n_vars, m, comps):
    """The oracle's restatement of BivariateMLEcheckProver (v3/bivariate_mlecheck.rs) passes the
    checks the reference's own test applies (compute_test_utils bivariate_sumcheck.rs:313-458):
    every round polynomial sums to the running claim, and the final values are the MLE evaluations
    at the (reversed) challenges with prod * eq == last claim."""
    mls, eq_ch, eq, sums = _mlecheck_instance(oracle, n_vars, m, comps, 0x3C3C00)
    stream = oracle.random_scalars(0xC4A2, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    coeffs, finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq.copy(), eq_ch, comps, sums, bc, ch)
    running = oracle.evaluate_univariate(sums, bc)
    for r, c in enumerate(coeffs):
        p0, p1 = c[0], c[0] ^ c[1] ^ c[2] ^ c[3]
        assert p0 ^ p1 == running, f"round {r}"
        running = oracle.evaluate_univariate(c, ch[r])
    point = list(reversed(ch))  # High-to-Low binding
    for x, f in zip(mls, finals[:m]):
        assert oracle.mle_evaluate(x, n_vars, point) == f
    # eq_ind_prefix_eval = eq(eq_ch, point) = prod (a_i + z_i + 1)
    want_eq = 1
    for a, z in zip(eq_ch, point):
        want_eq = oracle.mul(want_eq, a ^ z ^ 1)
    assert finals[m] == want_eq
    comp = 0
    p = 1
    for i, j in comps:
        comp ^= oracle.mul(p, oracle.mul(finals[i], finals[j]))
        p = oracle.mul(p, bc)
    assert oracle.mul(comp, finals[m]) == running
