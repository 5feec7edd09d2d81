"""GPU parity tests of the device-resident FRI commit + fold phases (the C++ mirror binius_amd/host/fri.hpp behind
bnh_fri_commit_fold; its query phase and error cases are checked in tests/cpp/conformance.cpp) against the
composition of the oracle's restatements (additive NTT, fold_interleaved / fri_fold, Groestl Merkle
tree), bit-exact.  Shapes follow crates/core/src/protocols/fri/tests.rs (test_commit_prove_verify_*:
log_dimension 8, log_inv_rate 2, log_batch_size 0/3, arities [3, 2, 1] / [4, 4] / no arities)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, 1 << 22)
    yield ctx
    ctx.close()


def ints(a):
    return [int(a[i, 0]) | (int(a[i, 1]) << 64) for i in range(a.shape[0])]


def oracle_commit(oracle, s_ref, log_domain, p, message):
    """commit_interleaved restated with the oracle's pieces."""
    code = np.concatenate([message] * (1 << p.log_inv_rate))
    assert oracle.ntt_forward(code, 5, 5, s_ref, log_domain, p.log_batch_size + 2, p.rs_log_len(), 0, 0, 0, p.log_inv_rate) == 0
    coset_log_len = p.fold_arities[0] if p.fold_arities else p.log_dim + p.log_batch_size
    rc, nodes = oracle.merkle_build(code, 1 << coset_log_len)
    assert rc == 0
    return code, nodes


@pytest.mark.parametrize(
    "log_dim,log_inv_rate,log_batch,arities",
    [(8, 2, 0, [3, 2, 1]), (8, 2, 3, [3, 2, 1]), (8, 2, 3, [4, 4]), (6, 1, 2, []), (10, 1, 4, [4, 4, 2]), (14, 1, 2, [4, 4, 4])],
)
def test_compiled_fri_matches_oracle(hal, oracle, log_dim, log_inv_rate, log_batch, arities):
    """bnh_fri_commit_fold (the C++ mirror behind one C call): every root and the terminal codeword against
    the oracle composition."""
    from binius_amd._host import FRIParams, FriPlan

    p = FRIParams(log_dim, log_inv_rate, log_batch, arities, n_test_queries=3)
    log_domain = p.rs_log_len()
    s_ref = oracle.ntt_s_evals(5, log_domain)
    alloc = hal.dev_alloc()
    message = oracle.random_b128(0xF77 + log_dim, 1 << (log_dim + log_batch))
    d_msg = alloc.alloc(message.shape[0])
    hal.copy_h2d(message, d_msg)
    scratch = alloc.alloc(2 << (log_dim + log_batch + log_inv_rate))
    challenges = oracle.random_scalars(0xC4B + log_dim, p.n_fold_rounds())
    plan = FriPlan(hal, p, d_msg, scratch, challenges)
    plan.run()
    code, nodes = oracle_commit(oracle, s_ref, log_domain, p, message)
    want_roots = [bytes(nodes[-1])]
    cur, cur_log_len, cur_log_batch, pos = code, p.rs_log_len(), p.log_batch_size, 0
    for k, arity in enumerate(arities):
        chs = challenges[pos : pos + arity]
        pos += arity
        new_log_len = cur_log_len - (len(chs) - cur_log_batch)
        nxt = oracle.arr(1 << new_log_len)
        assert oracle.fri_fold(s_ref, 5, log_domain, cur_log_len, cur_log_batch, chs, cur, nxt) == 0
        coset = 1 << (arities[k + 1] if k + 1 < len(arities) else p.n_final_challenges())
        rc, nd = oracle.merkle_build(nxt, coset)
        assert rc == 0
        want_roots.append(bytes(nd[-1]))
        cur, cur_log_len, cur_log_batch = nxt, new_log_len, 0
    assert [bytes(r) for r in plan.roots] == want_roots
    if arities:
        assert np.array_equal(plan.terminate, cur)
