"""The optimized CPU sumcheck (oracle/fastcpu_ref.c: POLYVAL-basis arithmetic with PCLMULQDQ, the
cpu_baseline leg of bench.py) produces the scalar restatement's transcript, bit for bit."""
import pytest


@pytest.fixture(scope="module")
def oracle():
    import oracle as o

    o.build()
    return o


@pytest.mark.parametrize("n_vars,m,comps,threads", [(1, 2, [(0, 1)], 1), (6, 2, [(0, 1)], 1), (10, 3, [(0, 1), (2, 0), (1, 1)], 4), (13, 2, [(0, 1)], 8)])
def test_fast_cpu_sumcheck_matches_scalar_restatement(oracle, n_vars, m, comps, threads):
    mls = [oracle.random_b128(0xFA57 + 16 * n_vars + j, 1 << n_vars) for j in range(m)]
    sums = [oracle.inner_product(mls[i], 7, mls[j])[1] for i, j in comps]
    stream = oracle.random_scalars(0xC4A1 + n_vars, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    want = oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges)
    got = oracle.fast_bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges, threads=threads)
    if got is None:
        pytest.skip("host without PCLMULQDQ")
    assert got[0] == want[0]
    assert got[1] == want[1]


def test_fast_cpu_sumcheck_matches_scalar_restatement_n20_three_compositions(oracle):
    """The pin the n = 24 / n = 28 GPU parity tests lean on (tests/test_gpu_north_star.py): at 2^20 with three batched
    compositions over three multilinears the PCLMULQDQ port and the scalar tower-recursion restatement agree bit for
    bit -- round polynomials and final evaluations."""
    import os

    n_vars, m, comps = 20, 3, [(0, 1), (2, 0), (1, 2)]
    threads = min(8, os.cpu_count() or 1)
    mls = [oracle.random_b128(0xFA5720 + j, 1 << n_vars) for j in range(m)]
    sums = [oracle.fast_inner_product(mls[i], mls[j], threads) for i, j in comps]
    if sums[0] is None:
        pytest.skip("host without PCLMULQDQ")
    stream = oracle.random_scalars(0xC4A1 + n_vars, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    want = oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges, threads=threads)
    got = oracle.fast_bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges, threads=threads)
    assert got[0] == want[0]
    assert got[1] == want[1]
    # the verifier's equations on the transcript: P_r(0) + P_r(1) = running sum, final product = last sum
    running = oracle.evaluate_univariate(sums, batch_coeff)
    for r, (c0, c1, c2) in enumerate(got[0]):
        assert c0 ^ (c0 ^ c1 ^ c2) == running
        running = oracle.evaluate_univariate([c0, c1, c2], challenges[r])
    fin = got[1]
    expect, p = 0, 1
    for i, j in comps:
        expect ^= oracle.mul(oracle.mul(fin[i], fin[j]), p)
        p = oracle.mul(p, batch_coeff)
    assert expect == running


@pytest.mark.parametrize("n", [1, 5, 1000, (1 << 16) + 3])
def test_fast_inner_product_matches_scalar_restatement(oracle, n):
    a, b = oracle.random_b128(0x1A57 + n, n), oracle.random_b128(0x1B57 + n, n)
    a0, b0 = a.copy(), b.copy()
    got = oracle.fast_inner_product(a, b, threads=3)
    if got is None:
        pytest.skip("host without PCLMULQDQ")
    rc, want = oracle.inner_product(a, 7, b)
    assert rc == 0 and got == want
    assert (a == a0).all() and (b == b0).all()
