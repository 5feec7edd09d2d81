"""Pins oracle/piop_ref.py -- the CPU restatement of piop::commit / piop::prove (crates/core/src/piop/prove.rs:106-395) and of the
front-loaded batch prover (protocols/sumcheck/prove/front_loaded.rs:33-203) -- with the VERIFIER's equations (tests/piop_verify.py:
BatchVerifier, final evaluations = extensions at the reversed challenges, evaluate_piecewise_multilinear = FRI final value; the last
one ties merge_multilins' order and bit reversal, the RS encoding, every fri_fold and the sumcheck finals together), on the
reference's own PIOP test shapes (crates/core/tests/piop.rs) with its own parameter choice, and on further FRI shapes."""
import pytest

from piop_verify import REFERENCE_SUITE, Params, make_instance, optimal_params, verify_transcript


def run(oracle, n_varss, n_transparents, p, seed):
    from oracle import piop_ref

    committed, transparents, claims = make_instance(oracle, n_varss, n_transparents, seed)
    meta = piop_ref.CommitMeta.with_vars(n_varss)
    assert meta.total_vars == p.n_fold_rounds()
    n_sizes = len(set(n_varss))
    stream = oracle.random_scalars(0x7A0 + len(n_varss), n_sizes + meta.total_vars)
    batch_coeffs, challenges = stream[:n_sizes], stream[n_sizes:]
    commitment, items, evals, terminate = piop_ref.piop_prove(committed, transparents, claims, p, batch_coeffs, challenges)
    assert len(commitment) == 32
    verify_transcript(oracle, piop_ref, n_varss, committed, transparents, claims, p, batch_coeffs, challenges, items)


@pytest.mark.parametrize(
    "n_varss,n_transparents,fri",
    [([3, 3, 5, 6], 2, (5, 1, 2, [2, 2])), ([4, 4, 4], 1, (4, 2, 2, [3])), ([2, 5], 2, (6, 1, 0, [2, 1, 1])), ([5, 5], 3, (6, 1, 0, []))],
)
def test_piop_restatement_satisfies_the_verifier(oracle, n_varss, n_transparents, fri):
    run(oracle, n_varss, n_transparents, Params(*fri), 0x9109 + 131 * len(n_varss))


@pytest.mark.parametrize("n_varss,n_transparents,log_inv_rate", REFERENCE_SUITE)
def test_reference_piop_suite_on_the_restatement(oracle, n_varss, n_transparents, log_inv_rate):
    """crates/core/tests/piop.rs: one polynomial, no opening claims at all, one size, an extreme rate, the small and the
    standard mix -- with make_commit_params_with_optimal_arity's parameters."""
    from oracle import piop_ref

    meta = piop_ref.CommitMeta.with_vars(n_varss)
    run(oracle, n_varss, n_transparents, optimal_params(meta.total_vars, log_inv_rate), 0x51ED + sum(n_varss))


def test_commit_meta_total_vars():
    """crates/core/src/piop/tests.rs:6-12"""
    from oracle import piop_ref

    assert piop_ref.CommitMeta.with_vars([4, 4, 6, 7]).total_vars == 8
    assert piop_ref.CommitMeta.with_vars([4, 4, 6, 6, 6, 7]).total_vars == 9


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
