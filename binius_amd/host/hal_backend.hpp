// binius_amd/host/hal_backend.hpp -- C++ mirror of the reference's OLD hardware abstraction layer,
// `binius_hal::ComputationBackend` (crates/hal/src/backend.rs:35-84), over device-resident multilinears:
//
//   tensor_product_full_query        backend.rs:41-44   -> bn_tensor_expand
//   sumcheck_compute_round_evals     backend.rs:52-67   -> bn_hal_round_evals
//   sumcheck_fold_multilinears       backend.rs:69-78   -> bn_hal_fold_multilinear per multilinear
//   evaluate_partial_high            backend.rs:80-84   -> bn_fold_left
//
// Same names, argument meaning and error behaviour as the trait; the one difference a device backend forces is
// that `Vcs<P>` (the reference's host-dereferenceable vector) is a handle into device memory (FSlice), so values
// only cross PCIe as round evaluations.
#pragma once
#include <optional>

#include "compute_layer.hpp"

namespace binius_amd {

enum class EvaluationOrder : uint32_t { LowToHigh = BN_ORDER_LOW_TO_HIGH, HighToLow = BN_ORDER_HIGH_TO_LOW }; // crates/math/src/fold.rs

// SumcheckMultilinear (crates/hal/src/sumcheck_multilinear.rs:8-30)
struct SumcheckMultilinear {
	enum Kind { Transparent, Folded } kind = Folded;
	// Transparent: packed subfield values of a multilinear over n_vars_ml variables, not yet evaluated at any challenge
	SubfieldSlice multilinear{FSlice{}, 7};
	size_t n_vars_ml = 0;
	size_t switchover_round = 0; // rounds left before it is partially evaluated at the tensor query
	// Folded: large-field evaluations after the challenges so far; the cube beyond them equals suffix_eval
	FSlice large_field_folded_evals{};
	B128 suffix_eval{};

	static SumcheckMultilinear transparent(SubfieldSlice ml, size_t n_vars_ml, size_t switchover_round)
	{
		SumcheckMultilinear m;
		m.kind = Transparent;
		m.multilinear = ml;
		m.n_vars_ml = n_vars_ml;
		m.switchover_round = switchover_round;
		return m;
	}
	static SumcheckMultilinear folded(FSlice evals, B128 suffix_eval = B128())
	{
		SumcheckMultilinear m;
		m.kind = Folded;
		m.large_field_folded_evals = evals;
		m.suffix_eval = suffix_eval;
		return m;
	}
	bn_hal_multilinear raw() const
	{
		bn_hal_multilinear r{};
		if (kind == Folded) {
			r.kind = BN_HAL_ML_FOLDED;
			r.d_evals = large_field_folded_evals.ptr;
			r.len = large_field_folded_evals.len_;
			r.suffix_eval = suffix_eval.raw();
		} else {
			r.kind = BN_HAL_ML_TRANSPARENT;
			r.tower_level = (uint32_t)multilinear.tower_level;
			r.d_evals = multilinear.slice.ptr;
			r.len = multilinear.slice.len_;
			r.n_vars_ml = (uint32_t)n_vars_ml;
		}
		return r;
	}
};

// What a SumcheckEvaluator (crates/hal/src/sumcheck_evaluator.rs:16-77) contributes to a round
struct SumcheckEvaluator {
	ExprEval composition;
	ExprEval composition_at_infinity; // ArithCircuit::leading_term (regular_sumcheck.rs:199-200)
	size_t eval_point_start = 0, eval_point_end = 0; // eval_point_indices
	std::optional<FSlice> eq_ind_partial_evals;      // EqIndSumcheckEvaluator (eq_ind.rs:676-704)
};

struct RoundEvals { // crates/hal/src/common.rs
	std::vector<B128> evals;
};

class Mi355xBackend {
public:
	explicit Mi355xBackend(ComputeLayer &hal) : hal_(hal) {}

	// backend.rs:41-44: the 2^len(query) tensor product expansion, left in device memory
	FSlice tensor_product_full_query(const std::vector<B128> &query, DeviceBumpAllocator &alloc)
	{
		FSliceMut buf = alloc.alloc((size_t)1 << query.size());
		FSliceMut first = ComputeMemory::slice_power_of_two_mut(buf, 1);
		hal_.fill(first, B128::ONE());
		hal_.execute([&](ComputeLayerExecutor &exec) {
			exec.tensor_expand(0, query, buf);
			return std::vector<B128>{};
		});
		return ComputeMemory::as_const(buf);
	}

	// backend.rs:52-67
	std::vector<RoundEvals> sumcheck_compute_round_evals(EvaluationOrder evaluation_order, size_t n_vars, std::optional<FSlice> tensor_query,
	                                                     const std::vector<SumcheckMultilinear> &multilinears,
	                                                     const std::vector<SumcheckEvaluator> &evaluators,
	                                                     const std::vector<B128> &nontrivial_evaluation_points)
	{
		std::vector<bn_hal_multilinear> mls;
		for (const auto &m : multilinears) mls.push_back(m.raw());
		std::vector<bn_hal_evaluator> evs;
		size_t total = 0;
		for (const auto &e : evaluators) {
			bn_hal_evaluator r{};
			r.composition = e.composition.handle();
			r.composition_at_infinity = e.composition_at_infinity.handle();
			r.eval_point_start = (uint32_t)e.eval_point_start;
			r.eval_point_end = (uint32_t)e.eval_point_end;
			r.d_eq_ind = e.eq_ind_partial_evals ? e.eq_ind_partial_evals->ptr : nullptr;
			evs.push_back(r);
			total += e.eval_point_end > e.eval_point_start ? e.eval_point_end - e.eval_point_start : 0;
		}
		std::vector<B128> flat(total ? total : 1);
		check(bn_hal_round_evals(hal_.raw_ctx(), (uint32_t)evaluation_order, (uint32_t)n_vars, tensor_query ? tensor_query->ptr : nullptr,
		                         tensor_query ? log2_exact(tensor_query->len_) : 0, mls.data(), (uint32_t)mls.size(), evs.data(), (uint32_t)evs.size(),
		                         reinterpret_cast<const bn_f128 *>(nontrivial_evaluation_points.data()), (uint32_t)nontrivial_evaluation_points.size(),
		                         reinterpret_cast<bn_f128 *>(flat.data())));
		std::vector<RoundEvals> out;
		size_t off = 0;
		for (const auto &e : evaluators) {
			const size_t cnt = e.eval_point_end > e.eval_point_start ? e.eval_point_end - e.eval_point_start : 0;
			out.push_back(RoundEvals{std::vector<B128>(flat.begin() + off, flat.begin() + off + cnt)});
			off += cnt;
		}
		return out;
	}

	// backend.rs:69-78: returns whether any multilinear is still transparent.  Folded multilinears are folded into
	// fresh device memory in Low-to-High order and in place in High-to-Low order (sumcheck_folding.rs:114-143, 218-232).
	bool sumcheck_fold_multilinears(EvaluationOrder evaluation_order, size_t n_vars, std::vector<SumcheckMultilinear> &multilinears, B128 challenge,
	                                std::optional<FSlice> tensor_query, DeviceBumpAllocator &alloc)
	{
		bool any_transparent_left = false;
		const bn_f128 z = challenge.raw();
		// High-to-Low order, every multilinear Folded and full: the fold is evals_0 += z (evals_1 - evals_0) on the halves, in place
		// (sumcheck_folding.rs:218-232 = extrapolate_line, compute/src/layer.rs:421) -- ALL multilinears of the call as one batch of the
		// ComputeLayer instead of a launch per multilinear (a constraint set's zerocheck folds every column of its table each round)
		if (evaluation_order == EvaluationOrder::HighToLow && n_vars >= 1) {
			bool plain = !multilinears.empty();
			for (const auto &m : multilinears)
				plain = plain && m.kind == SumcheckMultilinear::Folded && m.large_field_folded_evals.len_ == (size_t)1 << n_vars;
			if (plain) {
				struct Args {
					FSliceMut evals_0;
					FSlice evals_1;
				};
				std::vector<Args> prepared;
				for (auto &m : multilinears) {
					auto halves = ComputeMemory::split_half_mut(FSliceMut{const_cast<void *>(m.large_field_folded_evals.ptr), m.large_field_folded_evals.len_});
					prepared.push_back(Args{halves.first, ComputeMemory::to_const(halves.second)});
				}
				hal_.execute([&](ComputeLayerExecutor &exec) {
					exec.map(prepared.begin(), prepared.end(), [&](ComputeLayerExecutor &e, Args &a) {
						e.extrapolate_line(a.evals_0, a.evals_1, challenge);
						return 0;
					});
					return std::vector<B128>{};
				});
				for (size_t i = 0; i < multilinears.size(); i++)
					multilinears[i] = SumcheckMultilinear::folded(FSlice{prepared[i].evals_0.ptr, prepared[i].evals_0.len_}, multilinears[i].suffix_eval);
				return false;
			}
		}
		for (auto &m : multilinears) {
			if (m.kind == SumcheckMultilinear::Transparent && m.switchover_round > 0) {
				m.switchover_round--;
				any_transparent_left = true;
				continue;
			}
			const bn_hal_multilinear raw = m.raw();
			uint64_t n_out = 0;
			FSliceMut out{};
			if (m.kind == SumcheckMultilinear::Folded && evaluation_order == EvaluationOrder::HighToLow) {
				out = FSliceMut{const_cast<void *>(m.large_field_folded_evals.ptr), m.large_field_folded_evals.len_};
			} else {
				const size_t cap = m.kind == SumcheckMultilinear::Folded ? (m.large_field_folded_evals.len_ + 1) / 2 : (size_t)1 << (n_vars - 1);
				out = alloc.alloc(cap ? cap : 1);
			}
			if (m.kind == SumcheckMultilinear::Transparent && !tensor_query)
				throw Error(Error::InputValidation, "tensor query missing while a multilinear is still transparent");
			check(bn_hal_fold_multilinear(hal_.raw_ctx(), (uint32_t)evaluation_order, (uint32_t)n_vars, &raw, &z, tensor_query ? tensor_query->ptr : nullptr,
			                              tensor_query ? log2_exact(tensor_query->len_) : 0, out.ptr, out.len_, &n_out));
			m = SumcheckMultilinear::folded(FSlice{out.ptr, (size_t)n_out}, m.kind == SumcheckMultilinear::Folded ? m.suffix_eval : B128());
		}
		return any_transparent_left;
	}

	// backend.rs:80-84: partial evaluation of the high variables at the query expansion
	void evaluate_partial_high(const SubfieldSlice &multilinear, FSlice query_expansion, FSliceMut &out)
	{
		hal_.execute([&](ComputeLayerExecutor &exec) {
			exec.fold_left(multilinear, query_expansion, out);
			return std::vector<B128>{};
		});
	}

private:
	static uint32_t log2_exact(size_t n)
	{
		uint32_t l = 0;
		while (((size_t)1 << l) < n) l++;
		if (((size_t)1 << l) != n) throw Error(Error::InputValidation, "query expansion length must be a power of two");
		return l;
	}
	ComputeLayer &hal_;
};

// The caller of the backend: ProverState (crates/core/src/protocols/sumcheck/prove/prover_state.rs:57-265), the part
// of it that drives the multilinears through the rounds (the interpolation of round evaluations into coefficients is
// protocol-side scalar work and stays with the caller).
class ProverState {
public:
	ProverState(Mi355xBackend &backend, DeviceBumpAllocator &alloc, EvaluationOrder evaluation_order, size_t n_vars,
	            std::vector<SumcheckMultilinear> multilinears, std::vector<B128> nontrivial_evaluation_points)
	    : backend_(backend), alloc_(alloc), order_(evaluation_order), n_vars_(n_vars), multilinears_(std::move(multilinears)),
	      nontrivial_evaluation_points_(std::move(nontrivial_evaluation_points))
	{
		for (const auto &m : multilinears_) {
			if (m.kind == SumcheckMultilinear::Transparent) {
				if (m.n_vars_ml != n_vars) throw Error(Error::InputValidation, "NumberOfVariablesMismatch"); // (:94-96)
				has_query_ = true;
			}
		}
	}
	size_t n_vars() const { return n_vars_; }
	const std::vector<SumcheckMultilinear> &multilinears() const { return multilinears_; }

	// prover_state.rs:240-265
	std::vector<RoundEvals> calculate_round_evals(const std::vector<SumcheckEvaluator> &evaluators)
	{
		return backend_.sumcheck_compute_round_evals(order_, n_vars_, query(), multilinears_, evaluators, nontrivial_evaluation_points_);
	}
	// prover_state.rs:138-188
	void fold(B128 challenge)
	{
		if (n_vars_ == 0) throw Error(Error::InputValidation, "ExpectedFinish");
		if (order_ == EvaluationOrder::LowToHigh)
			challenges_.push_back(challenge);
		else
			challenges_.insert(challenges_.begin(), challenge);
		if (has_query_) tensor_query_ = backend_.tensor_product_full_query(challenges_, alloc_);
		const bool any_transparent_left = backend_.sumcheck_fold_multilinears(order_, n_vars_, multilinears_, challenge, query(), alloc_);
		if (!any_transparent_left) has_query_ = false;
		n_vars_--;
	}
	// prover_state.rs:190-225 for multilinears that have been folded (first stored evaluation, or the suffix)
	std::vector<B128> finish(ComputeLayer &hal) const
	{
		if (n_vars_ != 0) throw Error(Error::InputValidation, "ExpectedFold");
		std::vector<B128> out;
		for (const auto &m : multilinears_)
			if (m.kind != SumcheckMultilinear::Folded) throw Error(Error::InputValidation, "multilinear still transparent at finish");
		// (a table's worth of multilinears: their first evaluations in ONE gather -- one kernel, one synchronisation -- instead of a
		// copy_d2h and a synchronisation each; addresses relative to the lowest one)
		if (multilinears_.size() > 4) {
			const char *base = nullptr;
			for (const auto &m : multilinears_)
				if (m.large_field_folded_evals.len_ && (!base || (const char *)m.large_field_folded_evals.ptr < base)) base = (const char *)m.large_field_folded_evals.ptr;
			if (base) {
				std::vector<uint64_t> offs;
				for (const auto &m : multilinears_)
					if (m.large_field_folded_evals.len_) offs.push_back((uint64_t)((const char *)m.large_field_folded_evals.ptr - base) / sizeof(B128));
				std::vector<B128> vals(offs.size());
				check(bn_gather_d2h(hal.raw_ctx(), base, offs.data(), offs.size(), 1, reinterpret_cast<bn_f128 *>(vals.data())));
				size_t at = 0;
				for (const auto &m : multilinears_) out.push_back(m.large_field_folded_evals.len_ ? vals[at++] : m.suffix_eval);
				return out;
			}
		}
		for (const auto &m : multilinears_) {
			if (m.large_field_folded_evals.len_ == 0) {
				out.push_back(m.suffix_eval);
			} else {
				std::vector<B128> v(1);
				hal.copy_d2h(FSlice{m.large_field_folded_evals.ptr, 1}, v);
				out.push_back(v[0]);
			}
		}
		return out;
	}

private:
	std::optional<FSlice> query() const { return has_query_ && !challenges_.empty() ? std::optional<FSlice>(tensor_query_) : std::nullopt; }
	Mi355xBackend &backend_;
	DeviceBumpAllocator &alloc_;
	EvaluationOrder order_;
	size_t n_vars_;
	std::vector<SumcheckMultilinear> multilinears_;
	std::vector<B128> nontrivial_evaluation_points_;
	std::vector<B128> challenges_;
	FSlice tensor_query_{};
	bool has_query_ = false;
};

} // namespace binius_amd
