"""Independent mathematical pins for the parts of the oracle the reference holds no vectors for
(SURVEY.md section 8c): the restatements are checked against DEFINITIONS, not against
a second restatement by the same hand.

* normalized subspace polynomials (crates/ntt/src/twiddle.rs:236-244 doc comment): s_evals[i][b] must be
  W^_i(beta_{i+1+b}) with W_i(X) = prod_{u in U_i} (X - u), W^_i = W_i / W_i(beta_i), beta_k = 2^k
  (crates/field/src/binary_subspace.rs:33-38);
* forward additive NTT = evaluation of a polynomial given in the novel polynomial basis
  X_j = prod_{bit i of j} W^_i on the subspace points (crates/ntt/src/additive_ntt.rs:23-56, [DP24] 3.1),
  on the full domain S^(0) and on a smaller domain S^(l-k) whose basis is W^_{l-k}(beta_{l-k..});
* tensor_prod_eq_ind closed forms (crates/math/src/tensor_prod_eq_ind.rs:113-186): the expansion of
  (r_0, .., r_{k-1}) at index i is prod_k (bit_k(i) ? r_k : 1 - r_k)."""
import numpy as np
import pytest


def _mul(oracle, a, b, level):
    return oracle.gf_mul(a, b, level)


def _span(basis):
    pts = [0]
    for b in basis:
        pts += [p ^ b for p in pts]
    return pts


def _w_hat(oracle, basis, i, x, level):
    """W^_i(x) over U_i = span(basis[:i]), normalized at basis[i]."""
    def w(y):
        acc = 1
        for u in _span(basis[:i]):
            acc = _mul(oracle, acc, y ^ u, level)
        return acc
    return _mul(oracle, w(x), oracle.gf_invert(w(basis[i]), level), level)


@pytest.mark.parametrize("level,d", [(4, 7), (5, 8), (5, 5)])
def test_s_evals_are_normalized_subspace_polynomials(oracle, level, d):
    s = oracle.ntt_s_evals(level, d)
    stride = len(s) // int(round(len(s) ** 0.5))  # NTT_MAX_DIM x NTT_MAX_DIM
    beta = [1 << k for k in range(d)]
    for i in range(d):
        for b in range(d - 1 - i):
            assert int(s[i * stride + b]) == _w_hat(oracle, beta, i, beta[i + 1 + b], level), (i, b)
        # and W^_i(beta_i) = 1 is what the normalization means
        assert _w_hat(oracle, beta, i, beta[i], level) == 1


def _direct_novel_basis_eval(oracle, basis, coeffs, level):
    """evals[k] = sum_j coeffs[j] * X_j(omega_k), omega_k = sum_b bit_b(k) basis[b]."""
    d = len(basis)
    pts = []
    for k in range(1 << d):
        w = 0
        for b in range(d):
            if (k >> b) & 1:
                w ^= basis[b]
        pts.append(w)
    out = []
    for k in range(1 << d):
        wh = [_w_hat(oracle, basis, i, pts[k], level) for i in range(d)]
        acc = 0
        for j, c in enumerate(coeffs):
            x = c
            for i in range(d):
                if (j >> i) & 1:
                    x = _mul(oracle, x, wh[i], level)
            acc ^= x
        out.append(acc)
    return out


@pytest.mark.parametrize("level,log_domain,log_y", [(5, 5, 5), (5, 6, 6), (4, 6, 6), (5, 7, 5), (5, 6, 4)])
def test_forward_ntt_is_novel_basis_evaluation(oracle, level, log_domain, log_y):
    s = oracle.ntt_s_evals(level, log_domain)
    stride = len(s) // int(round(len(s) ** 0.5))
    n = 1 << log_y
    nbytes = 1 << (level - 3)
    rng = oracle.splitmix_words(0x4E5454 + log_y, n)
    coeffs = [int(x) & ((1 << (8 * nbytes)) - 1) for x in rng]
    data = np.array(coeffs, dtype={2: np.uint16, 4: np.uint32}[nbytes])
    assert oracle.ntt_forward(data, level, level, s, log_domain, 0, log_y, 0) == 0
    # domain S^(l-k): basis W^_{l-k}(beta_{l-k}), W^_{l-k}(beta_{l-k+1}), ... = (1, s_evals[l-k][0], ...)
    skip = log_domain - log_y
    basis = [1] + [int(s[skip * stride + b]) for b in range(log_y - 1)]
    if skip == 0:
        assert basis == [1 << k for k in range(log_y)]
    want = _direct_novel_basis_eval(oracle, basis, coeffs, level)
    assert [int(x) for x in data] == want
    # and the inverse transform interpolates back
    assert oracle.ntt_inverse(data, level, level, s, log_domain, 0, log_y, 0) == 0
    assert [int(x) for x in data] == coeffs


@pytest.mark.parametrize("k", [0, 1, 2, 3, 6])
def test_tensor_expand_closed_form(oracle, k):
    """math/src/tensor_prod_eq_ind.rs:113-186 (test_tensor_prod_eq_ind, test_eq_ind_partial_eval_*):
    entry i of the expansion is prod_j (bit_j(i) ? r_j : 1 - r_j) -- checked for the small constants the
    reference uses (1, 2, 3, 5 as field elements) and for random coordinates."""
    for coords in ([1, 2, 3, 5, 7, 11][:k], oracle.random_scalars(0x7E, k)):
        data = oracle.arr(1 << k)
        data[0] = (1, 0)
        oracle.tensor_expand(data, 0, coords)
        got = oracle.arr_to_ints(data)
        for i in range(1 << k):
            want = 1
            for j, r in enumerate(coords):
                want = oracle.mul(want, r if (i >> j) & 1 else (1 ^ r))
            assert got[i] == want, (k, i)
    if k == 0:
        data = oracle.arr(1)
        data[0] = (1, 0)
        oracle.tensor_expand(data, 0, [])
        assert oracle.arr_to_ints(data) == [1]  # test_eq_ind_partial_eval_empty
