// binius_amd/host/eq_ind.hpp -- C++ mirror of the caller of the OLD hardware abstraction layer that a constraint system's
// zerocheck runs on: EqIndSumcheckProver (crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644) over ProverState
// (prove/prover_state.rs:57-265; hal_backend.hpp), in the evaluation order High-to-Low.  A composition of degree d is asked for at the
// evaluation points 1 ..= d (eq_ind.rs:664-668): X = 1, infinity (d >= 2) and the points 2 .. d - 1 of the interpolation domain
// (d >= 3) -- DefaultEvaluationDomainFactory's: the first d elements 0, 1, 2, ... of the binary subspace plus infinity
// (math/src/univariate.rs:60-99; the nontrivial points, sumcheck/common.rs:310-340), and its prime polynomial is interpolated from
// (R'(0), R'(1), R'(2), ..., R'(inf)) (eq_ind.rs:753-779, univariate.rs:227-236: the Vandermonde system with the infinity row).  The
// tables SURVEY.md names (u32_add, keccak: m3/src/gadgets/hash/keccak/stacked.rs:142-151, 340-363) have constraints of degree 1 and
// 2; degree 3 and above take the old HAL's coefficient-form requests.
//
//   execute(batch_coeff)   eq_ind.rs:534-611   one evaluator per composition over ALL multilinears (sumcheck_compute_round_evals,
//                                              hal/src/backend.rs:52-67); per composition the "prime" round polynomial from (last sum,
//                                              R'(1), R'(inf)) (:753-779), batched (prover_state.rs:227-238), times eq(X, alpha) and
//                                              the prefix (:594-608)
//   fold(challenge)        eq_ind.rs:613-637   prefix *= eq(alpha, z); fold of every multilinear (sumcheck_fold_multilinears) and of
//                                              the indicator's partial evaluations (prove/common.rs:13-75: upper half onto lower)
//   finish()               eq_ind.rs:639-643   the multilinears' evaluations, then the prefix
//
// Protocol bookkeeping only: every hypercube-sized operation is a call of the backend.
#pragma once
#include "hal_backend.hpp"
#include "sumcheck.hpp"

namespace binius_amd {

struct EqIndComposition {
	ExprEval composition, composition_at_infinity; // the second = ArithCircuit::leading_term (eq_ind.rs:559-560)
	size_t degree = 2;                             // CompositionPoly::degree >= 1 (evaluation points 1 ..= degree, eq_ind.rs:664-668)
};

// the finite points of the interpolation domain: element i of the binary subspace = the tower element whose low bits are i
inline B128 eq_ind_domain_point(size_t i) { return B128((uint64_t)i, 0); }
constexpr size_t kEqIndMaxDegree = 8;

// InterpolationDomain::interpolate for a domain of the finite points 0 .. d - 1 and infinity (d >= 2), or 0, 1 (d = 1): the
// coefficients c_0 .. c_d of the polynomial of degree <= d with P(x_i) = finite[i] and leading coefficient `at_infinity`
// (univariate.rs:227-236, the matrix of :281-300 inverted) -- here: take c_d X^d off the finite values, Newton's divided
// differences through the d points, expanded to monomials
inline std::vector<B128> eq_ind_interpolate(const std::vector<B128> &finite, B128 at_infinity)
{
	const size_t d = finite.size();
	std::vector<B128> w = finite;
	if (d >= 2)
		for (size_t i = 0; i < d; i++) {
			B128 p = B128::ONE();
			for (size_t k = 0; k < d; k++) p = p * eq_ind_domain_point(i);
			w[i] = w[i] + at_infinity * p;
		}
	// divided differences (characteristic 2: x_i - x_j = x_i + x_j)
	std::vector<B128> dd = w;
	for (size_t level = 1; level < d; level++)
		for (size_t i = d - 1; i >= level; i--)
			dd[i] = (dd[i] + dd[i - 1]) * (eq_ind_domain_point(i) + eq_ind_domain_point(i - level)).invert_or_zero();
	// Horner over the Newton form: q = dd[d-1]; q = q (X - x_{i}) + dd[i]
	std::vector<B128> q(1, dd[d - 1]);
	for (size_t i = d - 1; i-- > 0;) {
		std::vector<B128> nq(q.size() + 1, B128::ZERO());
		for (size_t k = 0; k < q.size(); k++) {
			nq[k + 1] = nq[k + 1] + q[k];
			nq[k] = nq[k] + q[k] * eq_ind_domain_point(i);
		}
		nq[0] = nq[0] + dd[i];
		q.swap(nq);
	}
	q.resize(d >= 2 ? d + 1 : d, B128::ZERO());
	if (d >= 2) q[d] = at_infinity;
	return q;
}

class EqIndSumcheckProver {
public:
	// multilinears: Folded, full (2^n_vars elements), folded IN PLACE round by round (sumcheck_folding.rs:218-232);
	// eq_ind_partial_evals: 2^(n_vars - 1) elements = the tensor expansion of eq_ind_challenges[0 .. n_vars - 1) (eq_ind.rs:430-446),
	// folded in place too
	EqIndSumcheckProver(ComputeLayer &hal, Mi355xBackend &backend, DeviceBumpAllocator &alloc, size_t n_vars, std::vector<SumcheckMultilinear> multilinears,
	                    std::vector<EqIndComposition> compositions, std::vector<B128> sums, std::vector<B128> eq_ind_challenges, FSliceMut eq_ind_partial_evals)
	    : hal_(hal), n_vars_(n_vars), max_degree_(max_degree_of(compositions)),
	      state_(backend, alloc, EvaluationOrder::HighToLow, n_vars, std::move(multilinears), nontrivial_points(max_degree_)),
	      compositions_(std::move(compositions)), sums_(std::move(sums)), eq_ind_challenges_(std::move(eq_ind_challenges)), eq_ind_(eq_ind_partial_evals)
	{
		if (eq_ind_challenges_.size() != n_vars) throw SumcheckError("IncorrectEqIndChallengesLength");
		if (sums_.size() != compositions_.size()) throw SumcheckError("InvalidArgs(one sum per composition)");
		if (eq_ind_.len_ != (size_t)1 << (n_vars ? n_vars - 1 : 0)) throw SumcheckError("IncorrectEqIndPartialEvalsSize");
	}
	size_t n_vars() const { return n_vars_; }
	// coefficients of a round polynomial: the largest degree (at least 2) + 1 for the indicator's factor + 1
	size_t coeffs_per_round() const { return max_degree_ + 2; }

	// the round polynomial of degree max_degree + 1, batched over the compositions: coefficients c_0 .. c_{max_degree + 1}
	std::vector<B128> execute(B128 batch_coeff)
	{
		if (have_coeffs_) throw SumcheckError("ExpectedFold");
		const B128 alpha = eq_ind_round_challenge();
		std::vector<SumcheckEvaluator> evaluators;
		const FSlice eq{eq_ind_.ptr, eq_ind_.len_};
		for (const auto &c : compositions_) {
			SumcheckEvaluator e;
			e.composition = c.composition;
			e.composition_at_infinity = c.composition_at_infinity;
			if (c.degree < 1 || c.degree > kEqIndMaxDegree) throw SumcheckError("InvalidArgs(composition degree out of range)");
			e.eval_point_start = 1; // (:664-668: 1 ..= degree)
			e.eval_point_end = 1 + c.degree;
			e.eq_ind_partial_evals = eq;
			evaluators.push_back(e);
		}
		const std::vector<RoundEvals> round_evals = state_.calculate_round_evals(evaluators);
		// per composition: R'(0) = (sum - alpha R'(1)) / (1 - alpha), then c_0 = R'(0), c_2 = R'(inf), c_1 = R'(1) - c_0 - c_2 (:753-779)
		const B128 denom_inv = (B128::ONE() + alpha).invert_or_zero();
		const size_t n_prime = max_degree_ + 1;
		prime_coeffs_.assign(compositions_.size(), std::vector<B128>(n_prime));
		std::vector<B128> batched(n_prime, B128::ZERO());
		B128 scale = B128::ONE();
		for (size_t c = 0; c < compositions_.size(); c++) {
			// the evaluations come back in the order of the evaluation points 1, infinity (degree >= 2), domain points 2 .. degree - 1
			// (degree 1: the prime polynomial is linear -- no evaluation at infinity is asked for, its leading coefficient is zero)
			const size_t d = compositions_[c].degree;
			const B128 y1 = round_evals[c].evals[0], yinf = d >= 2 ? round_evals[c].evals[1] : B128::ZERO();
			const B128 y0 = (sums_[c] + y1 * alpha) * denom_inv;
			if (d <= 2) {
				prime_coeffs_[c].assign(n_prime, B128::ZERO());
				prime_coeffs_[c][0] = y0;
				prime_coeffs_[c][1] = y1 + y0 + yinf;
				prime_coeffs_[c][2] = yinf;
			} else {
				std::vector<B128> finite{y0, y1};
				for (size_t i = 2; i < d; i++) finite.push_back(round_evals[c].evals[i]);
				prime_coeffs_[c] = eq_ind_interpolate(finite, yinf);
				prime_coeffs_[c].resize(n_prime, B128::ZERO());
			}
			for (size_t i = 0; i < n_prime; i++) batched[i] = batched[i] + prime_coeffs_[c][i] * scale;
			scale = scale * batch_coeff;
		}
		have_coeffs_ = true;
		// v(X) = eq(X, alpha) v'(X) prefix = ((1 + alpha) + X) v'(X) prefix in characteristic 2 (:594-608)
		std::vector<B128> coeffs(n_prime + 1, B128::ZERO());
		for (size_t i = 0; i < n_prime; i++) {
			coeffs[i] = coeffs[i] + batched[i] * (B128::ONE() + alpha);
			coeffs[i + 1] = coeffs[i + 1] + batched[i];
		}
		for (auto &v : coeffs) v = v * eq_ind_prefix_eval_;
		return coeffs;
	}

	void fold(B128 challenge)
	{
		if (!have_coeffs_) throw SumcheckError("ExpectedExecution");
		eq_ind_prefix_eval_ = eq_ind_prefix_eval_ * eq(eq_ind_round_challenge(), challenge); // (:424-427)
		// the sums of the next round: every composition's prime polynomial at the challenge (prover_state.rs:150-157)
		for (size_t c = 0; c < compositions_.size(); c++) sums_[c] = evaluate_univariate(prime_coeffs_[c], challenge);
		have_coeffs_ = false;
		const size_t n_rounds_remaining = state_.n_vars();
		state_.fold(challenge);
		if (n_rounds_remaining - 1 > 0) fold_partial_eq_ind(n_rounds_remaining - 1);
	}

	std::vector<B128> finish()
	{
		std::vector<B128> evals = state_.finish(hal_);
		evals.push_back(eq_ind_prefix_eval_);
		return evals;
	}

private:
	static size_t max_degree_of(const std::vector<EqIndComposition> &cs)
	{
		size_t d = 2;
		for (const auto &c : cs) d = c.degree > d ? c.degree : d;
		if (d > kEqIndMaxDegree) throw SumcheckError("InvalidArgs(composition degree out of range)");
		return d;
	}
	// get_nontrivial_evaluation_points (sumcheck/common.rs:310-340): the finite points of the largest domain beyond 0 and 1
	static std::vector<B128> nontrivial_points(size_t max_degree)
	{
		std::vector<B128> p;
		for (size_t i = 2; i < max_degree; i++) p.push_back(eq_ind_domain_point(i));
		return p;
	}
	size_t round() const { return n_vars_ - state_.n_vars(); }
	B128 eq_ind_round_challenge() const { return eq_ind_challenges_[eq_ind_challenges_.size() - 1 - round()]; } // High-to-Low (:415-422)
	// prove/common.rs:13-75, High-to-Low: new[i] = old[i] + old[i | 2^(n_vars - 1)] -- the upper half onto the lower half
	void fold_partial_eq_ind(size_t n_vars)
	{
		auto halves = ComputeMemory::split_half_mut(eq_ind_);
		FSliceMut evals_0 = halves.first;
		const FSlice evals_1 = ComputeMemory::to_const(halves.second);
		std::vector<KernelMemMap> kernel_mappings{KernelMemMap::chunked_mut(evals_0, 0), KernelMemMap::chunked(evals_1, 0)};
		const size_t split_n_vars = n_vars - 1;
		hal_.execute([&](ComputeLayerExecutor &exec) {
			exec.map_kernels(
			    [&](KernelExecutor &local_exec, size_t log_chunks, std::vector<KernelBuffer> &buffers) {
				    local_exec.add_assign(split_n_vars - log_chunks, buffers[1].to_ref(), buffers[0].as_mut());
			    },
			    kernel_mappings);
			return std::vector<B128>{};
		});
		eq_ind_ = evals_0;
	}

	ComputeLayer &hal_;
	size_t n_vars_;
	size_t max_degree_; // of the compositions, at least 2
	ProverState state_;
	std::vector<EqIndComposition> compositions_;
	std::vector<B128> sums_; // per composition: the claimed sum, then its prime polynomial at the challenges so far
	std::vector<std::vector<B128>> prime_coeffs_;
	std::vector<B128> eq_ind_challenges_;
	FSliceMut eq_ind_;
	B128 eq_ind_prefix_eval_ = B128::ONE();
	bool have_coeffs_ = false;
};

} // namespace binius_amd
