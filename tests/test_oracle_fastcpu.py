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
