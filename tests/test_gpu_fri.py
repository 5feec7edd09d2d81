"""GPU parity tests of the device-resident FRI commit + fold phases (binius_amd/fri.py) against the
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
def test_commit_fold_query_matches_oracle(hal, oracle, log_dim, log_inv_rate, log_batch, arities):
    from binius_amd import fri
    from binius_amd.merkle import BinaryMerkleTreeProver

    p = fri.FRIParams(log_dim, log_inv_rate, log_batch, arities, n_test_queries=3)
    log_domain = p.rs_log_len()
    ntt = fri.AdditiveNTT(log_domain)
    s_ref = oracle.ntt_s_evals(5, log_domain)
    assert np.array_equal(np.asarray(ntt.s_evals), np.asarray(s_ref))
    alloc = hal.dev_alloc()
    message = oracle.random_b128(0xF21 + log_dim + 7 * log_batch, 1 << (log_dim + log_batch))
    d_msg = alloc.alloc(message.shape[0])
    hal.copy_h2d(message, d_msg)
    merkle = BinaryMerkleTreeProver(hal, alloc)

    # ---- commit phase
    out = fri.commit_interleaved(hal, alloc, p, ntt, merkle, d_msg)
    want_code, want_nodes = oracle_commit(oracle, s_ref, log_domain, p, message)
    assert np.array_equal(hal.copy_d2h(out.codeword), want_code)
    assert np.array_equal(out.committed.inner_nodes, want_nodes)
    assert out.commitment == bytes(want_nodes[-1])

    # ---- fold phase
    folder = fri.FRIFolder(hal, p, ntt, merkle, out.codeword, out.committed)
    challenges = oracle.random_scalars(0xC4A + log_dim, folder.n_rounds())
    want_rounds = []  # (codeword, nodes)
    pending, cur, cur_log_len, cur_log_batch = [], want_code, p.rs_log_len(), p.log_batch_size
    commit_rounds = list(np.cumsum(arities)) if arities else []
    for r, ch in enumerate(challenges, start=1):
        got_root = folder.execute_fold_round(alloc, ch)
        pending.append(ch)
        if r not in commit_rounds:
            assert got_root is None
            continue
        new_log_len = cur_log_len - (len(pending) - cur_log_batch)
        nxt = oracle.arr(1 << new_log_len)
        assert oracle.fri_fold(s_ref, 5, log_domain, cur_log_len, cur_log_batch, pending, cur, nxt) == 0
        k = len(want_rounds) + 1
        coset = 1 << (arities[k] if k < len(arities) else p.n_final_challenges())
        rc, nodes = oracle.merkle_build(nxt, coset)
        assert rc == 0
        want_rounds.append((nxt, nodes))
        assert got_root == bytes(nodes[-1])
        cur, cur_log_len, cur_log_batch, pending = nxt, new_log_len, 0, []
    assert len(folder.round_committed) == len(arities)
    for (d_code, tree), (w_code, w_nodes) in zip(folder.round_committed, want_rounds):
        assert np.array_equal(hal.copy_d2h(d_code), w_code)
        assert np.array_equal(tree.inner_nodes, w_nodes)
    with_early = fri.FRIFolder(hal, p, ntt, merkle, out.codeword, out.committed)
    with pytest.raises(fri.FriError, match="EarlyProverFinish"):
        with_early.finalize()
    terminate, qp = folder.finalize()
    assert np.array_equal(terminate, want_rounds[-1][0] if want_rounds else want_code)

    # ---- query phase: every opened coset equals the oracle's codeword slice, its Merkle branch leads to
    # the advertised layer digest, and (the verifier's fold check, fri/verify.rs) folding the opened coset
    # with that oracle's challenges gives the next oracle's value at the query position
    layers = qp.vcs_optimal_layers()
    depths = p.optimal_layer_depths()
    assert [len(l) for l in layers] == [1 << d for d in depths]
    codes = [want_code] + [c for c, _ in want_rounds]
    for index in {0, (1 << p.index_bits()) - 1, (0x5A5A5 % (1 << p.index_bits())) if p.index_bits() else 0}:
        openings = qp.prove_query(index)
        assert len(openings) == len(arities)
        idx = index
        for i, (values, branch) in enumerate(openings):
            arity = arities[i]
            if i > 0:
                idx >>= arity
            assert np.array_equal(values, codes[i][idx << arity : (idx + 1) << arity])
            log_n_cosets = (codes[i].shape[0].bit_length() - 1) - arity
            assert len(branch) == log_n_cosets - depths[i]
            leaf = oracle.groestl256(values.tobytes())
            top = oracle.merkle_root_from_branch(leaf, idx, branch)
            assert top == bytes(layers[i][idx >> (log_n_cosets - depths[i])])


def test_fri_params_and_errors(hal, oracle):
    from binius_amd import fri

    with pytest.raises(fri.FriError, match="InvalidFoldAritySequence"):
        fri.FRIParams(4, 1, 0, [2, 2], 1)
    p = fri.FRIParams(8, 2, 3, [3, 2, 1], 3)
    assert (p.n_fold_rounds(), p.n_oracles(), p.index_bits(), p.n_final_challenges(), p.log_len()) == (11, 3, 10, 5, 13)
    alloc = hal.dev_alloc()
    ntt = fri.AdditiveNTT(p.rs_log_len())
    from binius_amd.merkle import BinaryMerkleTreeProver

    with pytest.raises(fri.FriError, match="InvalidArgs"):
        fri.commit_interleaved(hal, alloc, p, ntt, BinaryMerkleTreeProver(hal, alloc), alloc.alloc(1 << 10))


@pytest.mark.parametrize("log_dim,log_inv_rate,log_batch,arities", [(8, 2, 3, [3, 2, 1]), (10, 1, 4, [4, 4, 2]), (6, 1, 2, [])])
def test_compiled_fri_matches_oracle(hal, oracle, log_dim, log_inv_rate, log_batch, arities):
    """bnh_fri_commit_fold (the C++ mirror behind one C call): every root and the terminal codeword against
    the oracle composition."""
    from binius_amd import fri
    from binius_amd._host import FriPlan

    p = fri.FRIParams(log_dim, log_inv_rate, log_batch, arities, n_test_queries=3)
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
