"""TEST INFRASTRUCTURE shared by tests/test_oracle_piop.py (CPU) and tests/test_gpu_group.py (GPU): the instance construction of
compute_test_utils/src/piop.rs:26-100 (commit_prove_verify), the FRI parameter choice of make_commit_params_with_optimal_arity
(piop/verify.rs:146-168, fri/common.rs:68-123, 199-250), and the VERIFIER's side of piop::verify restated independently of any
prover-side bookkeeping:

  * BatchVerifier (protocols/sumcheck/front_loaded.rs:56-230): the initial batched sum, RoundProof::recover
    (common.rs:176-182), evaluate at the challenge, a finished claim's batch-weighted composite evaluation leaves the sum
    (verify_sumcheck.rs:131-145), and the sum ends at zero;
  * a prover's final evaluations are its multilinears' extensions at the reversed challenges (High-to-Low binding);
  * piop::verify's last check (piop/verify.rs:343-358): evaluate_piecewise_multilinear (math/src/piecewise_multilinear.rs:46-117)
    of the committed evaluations at the challenges equals the FRI final value -- the repetition codeword the terminate codeword
    folds to under the final challenges (fri/verify.rs:152-223)."""
import math


class Params:
    """the FRIParams arithmetic (= binius_amd._host.FRIParams, which a CPU-only test must not import)"""

    def __init__(self, log_dim, log_inv_rate, log_batch_size, fold_arities, n_test_queries=3):
        self.log_dim, self.log_inv_rate, self.log_batch_size, self.fold_arities = log_dim, log_inv_rate, log_batch_size, list(fold_arities)
        self.n_test_queries = n_test_queries

    def rs_log_len(self):
        return self.log_dim + self.log_inv_rate

    def n_fold_rounds(self):
        return self.log_dim + self.log_batch_size

    def n_final_challenges(self):
        return self.n_fold_rounds() - sum(self.fold_arities)


def estimate_optimal_arity(log_block_length, digest_size=32, field_size=16):
    """fri/common.rs:224-250"""
    best, old = None, None
    for arity in range(1, log_block_length + 1):
        cost = ((log_block_length // 2) * digest_size + (1 << arity) * field_size) * (log_block_length - arity) // arity
        if old is not None and cost > old:
            break
        best, old = arity, cost
    return best if best is not None else 1


def optimal_params(total_vars, log_inv_rate, security_bits=32):
    """make_commit_params_with_optimal_arity (piop/verify.rs:146-168) -> FRIParams::choose_with_constant_fold_arity
    (fri/common.rs:68-123) with calculate_n_test_queries (:199-219), F = B128, digests of 32 bytes."""
    arity = estimate_optimal_arity(total_vars + log_inv_rate)
    log_dim, log_batch = max(0, total_vars - arity), min(total_vars, arity)
    field_size = 2.0 ** 128
    per_query_err = 0.5 * (1.0 + 2.0 ** (-log_inv_rate))
    allowed = 2.0 ** (-security_bits) - (2 * log_dim) / field_size - (2.0 ** (log_dim + log_inv_rate)) / field_size
    n_queries = int(math.ceil(math.log(allowed) / math.log(per_query_err)))
    cap_height = max(0, (n_queries - 1).bit_length())
    n_arities = max(0, total_vars - max(0, cap_height - log_inv_rate)) // arity
    return Params(log_dim, log_inv_rate, log_batch, [arity] * n_arities, n_queries)


def make_instance(oracle, n_varss, n_transparents, seed):
    """compute_test_utils/src/piop.rs:26-100: random committed multilinears, n_transparents random transparents per size that has
    a committed one, a claim for every (committed, transparent) pair of equal size with its true sum."""
    committed = [oracle.random_b128(seed + 16 * i, 1 << v) for i, v in enumerate(n_varss)]
    t_sizes = [v for v in sorted(set(n_varss)) for _ in range(n_transparents)]
    transparents = [oracle.random_b128(seed + 0x1000 + 16 * j, 1 << v) for j, v in enumerate(t_sizes)]
    claims = []
    for i, c in enumerate(committed):
        for j, t in enumerate(transparents):
            if c.shape[0] == t.shape[0]:
                rc, s = oracle.inner_product(c, 7, t)
                assert rc == 0
                claims.append((c.shape[0].bit_length() - 1, i, j, s))
    return committed, transparents, claims


def batch_weighted_value(oracle, bc, values):
    acc, p = 0, 1
    for v in values:
        acc ^= oracle.mul(p, v)
        p = oracle.mul(p, bc)
    return oracle.mul(bc, acc)


def piecewise(oracle, point, n_pieces_by_vars, evals):
    """math/src/piecewise_multilinear.rs:46-117"""
    evals = list(evals)
    index, n_to_fold = len(evals), 0

    def line(a, b, z):
        return a ^ oracle.mul(z, a ^ b)

    for i, z in enumerate(point):
        n_to_fold += n_pieces_by_vars[i] if i < len(n_pieces_by_vars) else 0
        seg = evals[index - n_to_fold : index]
        for q in range(len(seg) // 2):
            seg[q] = line(seg[2 * q], seg[2 * q + 1], z)
        if len(seg) % 2 == 1:
            seg[len(seg) // 2] = line(seg[-1], 0, z)
        evals[index - n_to_fold : index] = seg
        index -= n_to_fold // 2
        n_to_fold -= n_to_fold // 2
    return evals[0]


def verify_transcript(oracle, piop_ref, n_varss, committed, transparents, claims, p, batch_coeffs, challenges, items):
    """piop::verify's equations on a transcript (list of (kind, payload) in writing order); asserts."""
    meta = piop_ref.CommitMeta.with_vars(n_varss)
    sizes = [v for v in range(meta.max_n_vars() + 1) if meta.n_multilins_by_vars[v]]
    assert [k for k, _ in items].count("fri_commitment") == len(p.fold_arities)
    # ---- BatchVerifier over the transcript
    descs = piop_ref.make_sumcheck_claim_descs(meta, [t.shape[0].bit_length() - 1 for t in transparents], claims)
    live = [(v, descs[v], bc) for v, bc in zip(sizes, batch_coeffs)]
    total = 0
    for v, d, bc in live:
        total ^= batch_weighted_value(oracle, bc, d["sums"])
    rnd, finished, terminate = 0, [], None
    for kind, payload in items:
        if kind == "multilinear_evals":
            v, d, bc = live.pop(0)
            assert v == rnd, "a prover finishes in the round that equals its number of variables"
            assert len(payload) == (d["committed"][1] - d["committed"][0]) + (d["transparent"][1] - d["transparent"][0])
            total ^= batch_weighted_value(oracle, bc, [oracle.mul(payload[i], payload[j]) for i, j in d["comps"]])
            finished.append((v, d, payload))
        elif kind == "round_proof":
            degree = 0
            for _, d, _ in live:  # max_degree_remaining: the bivariate claims have degree 2, a prover without claims degree 0
                degree = max(degree, 2 if d["comps"] else 0)
            assert len(payload) == degree and (not live or live[0][0] != rnd)
            first = payload[0] if payload else 0
            last = total ^ first
            for c in payload:
                last ^= c
            total = oracle.evaluate_univariate(list(payload) + [last], challenges[rnd])
            rnd += 1
        elif kind == "fri_terminate":
            terminate = payload
    assert rnd == meta.total_vars and not live and total == 0
    # ---- final evaluations = multilinear extensions at the reversed challenges
    for v, d, payload in finished:
        cb, ce = d["committed"]
        tb, te = d["transparent"]
        mls = committed[cb:ce] + transparents[tb:te]
        point = list(reversed(challenges[:v]))
        for x, got in zip(mls, payload):
            assert got == (oracle.mle_evaluate(x, v, point) if v else oracle.arr_to_ints(x)[0])
    # ---- committed evaluations against the FRI final value (piop/verify.rs:343-358)
    piece_evals = []
    for v, d, payload in finished:
        piece_evals += payload[: d["committed"][1] - d["committed"][0]]
    piece_evals.reverse()
    want = piecewise(oracle, challenges, meta.n_multilins_by_vars, piece_evals)
    f = p.n_final_challenges()
    term = oracle.ints_to_arr(terminate)
    s_evals = oracle.ntt_s_evals(5, p.rs_log_len())
    rep = oracle.arr(1 << p.log_inv_rate)
    if p.fold_arities:
        if f:
            assert oracle.fri_fold(s_evals, 5, p.rs_log_len(), f + p.log_inv_rate, 0, challenges[meta.total_vars - f :], term, rep) == 0
        else:
            rep = term
    else:
        assert oracle.fri_fold(s_evals, 5, p.rs_log_len(), p.rs_log_len(), p.log_batch_size, challenges, term, rep) == 0
    rep = oracle.arr_to_ints(rep)
    assert all(x == rep[0] for x in rep), "the terminate codeword does not fold to a repetition codeword"
    assert rep[0] == want, "committed evaluations do not match the FRI final value"


# the reference's own PIOP tests (crates/core/tests/piop.rs:10-112): (CommitMeta::with_vars, n_transparents, log_inv_rate)
REFERENCE_SUITE = [([4], 1, 1), ([4, 4, 6, 7], 0, 1), ([4, 4], 1, 1), ([3, 3, 5, 6], 2, 8), ([4, 4, 6, 7], 2, 1), ([6, 6, 8, 9], 2, 1)]
