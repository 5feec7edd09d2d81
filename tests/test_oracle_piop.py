"""Pins oracle/piop_ref.py -- the CPU restatement of piop::commit / piop::prove (crates/core/src/piop/prove.rs:106-395) and of the
front-loaded batch prover (protocols/sumcheck/prove/front_loaded.rs:33-203) -- with the VERIFIER's equations, restated here
independently of the prover-side bookkeeping:

  * BatchVerifier (protocols/sumcheck/front_loaded.rs:56-230): the initial batched sum, RoundProof::recover
    (common.rs:176-182), evaluate at the challenge, a finished claim's batch-weighted composite evaluation leaves the sum
    (verify_sumcheck.rs:131-145), and the sum ends at zero;
  * a prover's final evaluations are its multilinears' extensions at the reversed challenges (High-to-Low binding);
  * piop::verify's last check (piop/verify.rs:343-358): evaluate_piecewise_multilinear (math/src/piecewise_multilinear.rs:46-117)
    of the committed evaluations at the challenges equals the FRI final value -- the repetition codeword the terminate codeword
    folds to under the final challenges (fri/verify.rs:152-223).  This ties merge_multilins' order and bit reversal, the RS
    encoding, every fri_fold and the sumcheck finals together.

Shapes follow compute_test_utils/src/piop.rs:101 (commit_prove_verify: every committed multilinear against every transparent of
its size) and the reference's own PIOP tests (CommitMeta::with_vars mixes such as [6, 6, 8, 9], crates/core/src/piop/tests.rs)."""
import numpy as np
import pytest


class Params:
    """the FRIParams arithmetic oracle/piop_ref.py needs (= binius_amd._host.FRIParams, which a CPU-only test must not import)"""

    def __init__(self, log_dim, log_inv_rate, log_batch_size, fold_arities):
        self.log_dim, self.log_inv_rate, self.log_batch_size, self.fold_arities = log_dim, log_inv_rate, log_batch_size, list(fold_arities)

    def rs_log_len(self):
        return self.log_dim + self.log_inv_rate

    def n_fold_rounds(self):
        return self.log_dim + self.log_batch_size

    def n_final_challenges(self):
        return self.n_fold_rounds() - sum(self.fold_arities)


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


@pytest.mark.parametrize(
    "n_varss,n_transparents,fri",
    [([3, 3, 5, 6], 2, (5, 1, 2, [2, 2])), ([4, 4, 4], 1, (4, 2, 2, [3])), ([2, 5], 2, (6, 1, 0, [2, 1, 1])), ([5, 5], 3, (6, 1, 0, []))],
)
def test_piop_restatement_satisfies_the_verifier(oracle, n_varss, n_transparents, fri):
    from oracle import piop_ref

    p = Params(*fri)
    committed, transparents, claims = make_instance(oracle, n_varss, n_transparents, 0x9109 + 131 * len(n_varss))
    meta = piop_ref.CommitMeta.with_vars(n_varss)
    assert meta.total_vars == p.n_fold_rounds()
    sizes = [v for v in range(meta.max_n_vars() + 1) if meta.n_multilins_by_vars[v]]
    stream = oracle.random_scalars(0x7A0 + len(n_varss), len(sizes) + meta.total_vars)
    batch_coeffs, challenges = stream[: len(sizes)], stream[len(sizes) :]
    commitment, items, evals, terminate = piop_ref.piop_prove(committed, transparents, claims, p, batch_coeffs, challenges)
    assert len(commitment) == 32 and [k for k, _ in items].count("fri_commitment") == len(p.fold_arities)

    # ---- BatchVerifier over the transcript
    descs = piop_ref.make_sumcheck_claim_descs(meta, [t.shape[0].bit_length() - 1 for t in transparents], claims)
    live = [(v, descs[v], bc) for v, bc in zip(sizes, batch_coeffs)]
    total = 0
    for v, d, bc in live:
        total ^= batch_weighted_value(oracle, bc, d["sums"])
    rnd, finished = 0, []
    for kind, payload in items:
        if kind == "multilinear_evals":
            v, d, bc = live.pop(0)
            assert v == rnd, "a prover finishes in the round that equals its number of variables"
            assert len(payload) == (d["committed"][1] - d["committed"][0]) + (d["transparent"][1] - d["transparent"][0])
            total ^= batch_weighted_value(oracle, bc, [oracle.mul(payload[i], payload[j]) for i, j in d["comps"]])
            finished.append((v, d, payload))
        elif kind == "round_proof":
            degree = 2 if live else 0
            assert len(payload) == degree and (not live or live[0][0] != rnd)
            first = payload[0] if payload else 0
            last = total ^ first
            for c in payload:
                last ^= c
            total = oracle.evaluate_univariate(list(payload) + [last], challenges[rnd])
            rnd += 1
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
    if p.fold_arities:
        s_evals = oracle.ntt_s_evals(5, p.rs_log_len())
        rep = oracle.arr(1 << p.log_inv_rate)
        if f:
            assert oracle.fri_fold(s_evals, 5, p.rs_log_len(), f + p.log_inv_rate, 0, challenges[meta.total_vars - f :], term, rep) == 0
        else:
            rep = term
    else:
        s_evals = oracle.ntt_s_evals(5, p.rs_log_len())
        rep = oracle.arr(1 << p.log_inv_rate)
        assert oracle.fri_fold(s_evals, 5, p.rs_log_len(), p.rs_log_len(), p.log_batch_size, challenges, term, rep) == 0
    rep = oracle.arr_to_ints(rep)
    assert all(x == rep[0] for x in rep), "the terminate codeword does not fold to a repetition codeword"
    assert rep[0] == want, "committed evaluations do not match the FRI final value"


def test_batch_prover_bookkeeping_small(oracle):
    """front_loaded.rs: a prover with no compositions contributes nothing to the round polynomials but still owes its final
    evaluations; a zero-variable prover finishes before the first round; the fast (PCLMULQDQ) per-prover oracle gives the
    scalar one's transcript."""
    from oracle import piop_ref

    def provers():
        return [
            dict(n_vars=0, multilins=[oracle.random_b128(1, 1), oracle.random_b128(2, 1)], comps=[(0, 1)], sums=[]),
            dict(n_vars=3, multilins=[oracle.random_b128(3 + j, 8) for j in range(3)], comps=[], sums=[]),
            dict(n_vars=5, multilins=[oracle.random_b128(7 + j, 32) for j in range(4)], comps=[(0, 2), (1, 3), (0, 3)], sums=[]),
        ]

    ps = provers()
    for p in ps:
        p["sums"] = [oracle.inner_product(p["multilins"][i], 7, p["multilins"][j])[1] for i, j in p["comps"]]
    stream = oracle.random_scalars(0xBA7C, 3 + 5)
    items, evals = piop_ref.batch_sumcheck_prove(ps, stream[:3], stream[3:])
    assert [k for k, _ in items] == ["multilinear_evals"] + ["round_proof"] * 3 + ["multilinear_evals"] + ["round_proof"] * 2 + ["multilinear_evals"]
    assert [len(e) for e in evals] == [2, 3, 4]
    ps2 = provers()
    for p, q in zip(ps2, ps):
        p["sums"] = q["sums"]
    items2, evals2 = piop_ref.batch_sumcheck_prove(ps2, stream[:3], stream[3:], fast=True)
    assert items2 == items and evals2 == evals
