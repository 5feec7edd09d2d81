// binius_amd/host/sumcheck.hpp -- C++ mirror of the reference's callers of the HAL on the measured
// path, generic over nothing but the ComputeLayer of compute_layer.hpp:
//
//   ops::eq_ind_partial_eval       crates/compute/src/ops.rs:26-50
//   calculate_round_evals          crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-408
//   (with eq_ind)                  crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:391-520
//   round coeffs from evals        v3/bivariate_product.rs:410-424
//   BivariateSumcheckProver        v3/bivariate_product.rs:27-254  (execute / fold / finish)
//   BivariateMLEcheckProver        v3/bivariate_mlecheck.rs:27-389  (execute / fold / finish, fold_eq_ind)
//   evaluate_univariate            crates/math/src/univariate.rs:264-270
//
// Protocol bookkeeping only; every hypercube-sized operation is a HAL call.
#pragma once

#include <stdexcept>
#include <vector>

#include "compute_layer.hpp"

namespace binius_amd {

inline B128 evaluate_univariate(const std::vector<B128> &coeffs, B128 x)
{
	B128 e = B128::ZERO();
	for (size_t i = coeffs.size(); i-- > 0;) e = e * x + coeffs[i];
	return e;
}

inline std::vector<B128> powers(B128 x, size_t n)
{
	std::vector<B128> out;
	B128 p = B128::ONE();
	for (size_t i = 0; i < n; i++) {
		out.push_back(p);
		p = p * x;
	}
	return out;
}

// IndexComposition<BivariateProduct, 2> {indices}: expression = var(i0) * var(i1)
// (core/src/composition/product_composition.rs:30-32, index.rs:50-55)
struct IndexCompositionBivariate {
	size_t n_vars;
	size_t indices[2];
	ArithCircuit expression() const { return (ArithCircuit::var(0) * ArithCircuit::var(1)).remap_vars({indices[0], indices[1]}); }
};

namespace ops {
// eq_ind_partial_eval (ops.rs:26-50)
inline FSliceMut eq_ind_partial_eval(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, const std::vector<B128> &point)
{
	const size_t n_vars = point.size();
	FSliceMut out = dev_alloc.alloc((size_t)1 << n_vars);
	{
		FSliceMut dev_val = ComputeMemory::slice_power_of_two_mut(out, 1);
		hal.fill(dev_val, B128::ONE());
	}
	hal.execute([&](ComputeLayerExecutor &exec) {
		exec.tensor_expand(0, point, out);
		return std::vector<B128>{};
	});
	return out;
}
} // namespace ops

// calculate_round_evals: returns {y_1, y_inf}.  eq_ind (optional) is appended as the last
// composition variable, as the MLE-check prover does.
inline std::vector<B128> calculate_round_evals(ComputeLayer &hal, size_t n_vars, B128 batch_coeff, const std::vector<FSlice> &multilins,
                                               const std::vector<ExprEval> &prod_evaluators, const FSlice *eq_ind = nullptr)
{
	const size_t split_n_vars = n_vars - 1;
	std::vector<KernelMemMap> kernel_mappings;
	for (const FSlice &ml : multilins) {
		auto halves = ComputeMemory::split_half(ml);
		kernel_mappings.push_back(KernelMemMap::chunked(halves.first, 0));
		kernel_mappings.push_back(KernelMemMap::chunked(halves.second, 0));
		kernel_mappings.push_back(KernelMemMap::local(split_n_vars)); // evaluations at the extra point
	}
	if (eq_ind) kernel_mappings.push_back(KernelMemMap::chunked(*eq_ind, 0));
	const std::vector<B128> batch_coeffs = powers(batch_coeff, prod_evaluators.size());
	const size_t m = multilins.size();

	return hal.execute([&](ComputeLayerExecutor &exec) {
		return exec.accumulate_kernels(
		    [&](KernelExecutor &local_exec, size_t log_chunks, std::vector<KernelBuffer> &buffers) {
			    const size_t log_chunk_size = split_n_vars - log_chunks;
			    // composite evaluations at ONE
			    KernelValue acc_1 = local_exec.decl_value(B128::ZERO());
			    {
				    std::vector<KSlice> rows;
				    for (size_t i = 0; i < m; i++) rows.push_back(buffers[i * 3 + 1].to_ref());
				    if (eq_ind) rows.push_back(buffers.back().to_ref());
				    SlicesBatch<KSlice> eval_1s(rows, (size_t)1 << log_chunk_size);
				    for (size_t c = 0; c < prod_evaluators.size(); c++)
					    local_exec.sum_composition_evals(eval_1s, prod_evaluators[c], batch_coeffs[c], acc_1);
			    }
			    // extrapolate to the point at infinity: evals_inf = evals_0 + evals_1
			    for (size_t i = 0; i < m; i++)
				    local_exec.add(log_chunk_size, buffers[3 * i].to_ref(), buffers[3 * i + 1].to_ref(), buffers[3 * i + 2].as_mut());
			    KernelValue acc_inf = local_exec.decl_value(B128::ZERO());
			    {
				    std::vector<KSlice> rows;
				    for (size_t i = 0; i < m; i++) rows.push_back(buffers[i * 3 + 2].to_ref());
				    if (eq_ind) rows.push_back(buffers.back().to_ref());
				    SlicesBatch<KSlice> eval_infs(rows, (size_t)1 << log_chunk_size);
				    for (size_t c = 0; c < prod_evaluators.size(); c++)
					    local_exec.sum_composition_evals(eval_infs, prod_evaluators[c], batch_coeffs[c], acc_inf);
			    }
			    return std::vector<KernelValue>{acc_1, acc_inf};
		    },
		    kernel_mappings);
	});
}

// RoundCoeffs c0, c1, c2 from (sum, y_1, y_inf) (v3/bivariate_product.rs:410-424)
inline std::vector<B128> calculate_round_coeffs_from_evals(B128 sum, const std::vector<B128> &evals)
{
	const B128 y_1 = evals[0], y_inf = evals[1];
	const B128 y_0 = sum - y_1;
	const B128 c_0 = y_0, c_2 = y_inf;
	const B128 c_1 = y_1 - c_0 - c_2;
	return {c_0, c_1, c_2};
}

class SumcheckError : public std::logic_error {
public:
	using std::logic_error::logic_error;
};

// BivariateSumcheckProver (v3/bivariate_product.rs:27-254); evaluation order High-to-Low.
class BivariateSumcheckProver {
public:
	BivariateSumcheckProver(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, HostBumpAllocator &host_alloc, size_t n_vars,
	                        const std::vector<IndexCompositionBivariate> &compositions, const std::vector<B128> &sums,
	                        const std::vector<FSlice> &multilins)
	    : hal_(hal), dev_alloc_(dev_alloc), host_alloc_(host_alloc), n_vars_initial_(n_vars), n_vars_remaining_(n_vars)
	{
		for (const auto &ml : multilins)
			if (ml.len() != (size_t)1 << n_vars) throw SumcheckError("NumberOfVariablesMismatch");
		for (const auto &ml : multilins) multilins_.push_back(Multilin{true, FSliceMut{const_cast<void *>(ml.ptr), ml.len_}});
		for (const auto &c : compositions) evaluators_.push_back(hal.compile_expr(c.expression()));
		state_ = InitialSums;
		sums_or_coeffs_ = sums;
	}
	static size_t required_host_memory(size_t n_multilinears) { return n_multilinears; }
	static size_t required_device_memory(size_t n_multilinears, size_t n_vars) { return n_multilinears * ((size_t)1 << (n_vars - 1)); }
	size_t n_vars() const { return n_vars_initial_; }

	std::vector<B128> execute(B128 batch_coeff)
	{
		std::vector<FSlice> mls;
		for (const auto &m : multilins_) mls.push_back(FSlice{m.evals.ptr, m.evals.len_});
		const std::vector<B128> round_evals = calculate_round_evals(hal_, n_vars_remaining_, batch_coeff, mls, evaluators_);
		B128 batched_sum;
		switch (state_) {
		case Coeffs: throw SumcheckError("ExpectedFold");
		case InitialSums: batched_sum = evaluate_univariate(sums_or_coeffs_, batch_coeff); break;
		default: batched_sum = batched_sum_; break;
		}
		std::vector<B128> round_coeffs = calculate_round_coeffs_from_evals(batched_sum, round_evals);
		state_ = Coeffs;
		sums_or_coeffs_ = round_coeffs;
		if (evaluators_.empty()) return {};
		return round_coeffs;
	}

	void fold(B128 challenge)
	{
		if (n_vars_remaining_ == 0) throw SumcheckError("ExpectedFinish");
		if (state_ != Coeffs) throw SumcheckError("ExpectedExecution");
		batched_sum_ = evaluate_univariate(sums_or_coeffs_, challenge);
		state_ = BatchedSum;
		struct Args {
			FSliceMut evals_0;
			FSlice evals_1;
		};
		std::vector<Args> prepared;
		for (auto &m : multilins_) {
			if (m.pre_fold) {
				auto halves = ComputeMemory::split_half(FSlice{m.evals.ptr, m.evals.len_});
				// allocate a new buffer for the folded evaluations and copy in evals_0
				FSliceMut folded = dev_alloc_.alloc((size_t)1 << (n_vars_remaining_ - 1));
				hal_.copy_d2d(halves.first, folded);
				prepared.push_back(Args{folded, halves.second});
			} else {
				auto halves = ComputeMemory::split_half_mut(m.evals);
				prepared.push_back(Args{halves.first, ComputeMemory::to_const(halves.second)});
			}
		}
		hal_.execute([&](ComputeLayerExecutor &exec) {
			auto folded = exec.map(prepared.begin(), prepared.end(), [&](ComputeLayerExecutor &e, Args &a) {
				e.extrapolate_line(a.evals_0, a.evals_1, challenge);
				return Multilin{false, a.evals_0};
			});
			multilins_ = folded;
			return std::vector<B128>{};
		});
		n_vars_remaining_ -= 1;
	}

	std::vector<B128> finish()
	{
		if (state_ == Coeffs) throw SumcheckError("ExpectedFold");
		if (n_vars_remaining_ != 0) throw SumcheckError("ExpectedExecution");
		HostSliceMut buffer = host_alloc_.alloc(multilins_.size());
		for (size_t i = 0; i < multilins_.size(); i++)
			hal_.copy_d2h(FSlice{multilins_[i].evals.ptr, multilins_[i].evals.len_}, &buffer[i], 1);
		return std::vector<B128>(buffer.ptr, buffer.ptr + multilins_.size());
	}

	struct Multilin {
		bool pre_fold;
		FSliceMut evals;
	};
	const std::vector<Multilin> &multilins() const { return multilins_; }

private:
	enum State { Coeffs, InitialSums, BatchedSum };
	ComputeLayer &hal_;
	DeviceBumpAllocator &dev_alloc_;
	HostBumpAllocator &host_alloc_;
	size_t n_vars_initial_, n_vars_remaining_;
	std::vector<Multilin> multilins_;
	std::vector<ExprEval> evaluators_;
	State state_;
	std::vector<B128> sums_or_coeffs_;
	B128 batched_sum_;
};

// eq(x, y): the 2-variate multilinear indicating x == y (field/src/util.rs:72-81); characteristic 2
inline B128 eq(B128 x, B128 y) { return x + y + B128::ONE(); }

// BivariateMLEcheckProver (v3/bivariate_mlecheck.rs:27-372): the eq-indicator sumcheck for bivariate
// products.  eq_ind_partial_evals = tensor expansion of eq_ind_challenges[0 .. n_vars-1).
class BivariateMLEcheckProver {
public:
	BivariateMLEcheckProver(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, HostBumpAllocator &host_alloc, size_t n_vars,
	                        const std::vector<IndexCompositionBivariate> &compositions, const std::vector<B128> &sums,
	                        const std::vector<FSlice> &multilins, FSlice eq_ind_partial_evals, std::vector<B128> eq_ind_challenges)
	    : hal_(hal), dev_alloc_(dev_alloc), host_alloc_(host_alloc), n_vars_initial_(n_vars), n_vars_remaining_(n_vars),
	      eq_ind_challenges_(std::move(eq_ind_challenges))
	{
		for (const auto &ml : multilins)
			if (ml.len() != (size_t)1 << n_vars) throw SumcheckError("NumberOfVariablesMismatch");
		// only one value of the expanded indicator is used per 1-variable subcube (:97-101)
		if (eq_ind_partial_evals.len() != (size_t)1 << (n_vars ? n_vars - 1 : 0)) throw SumcheckError("IncorrectEqIndPartialEvalsSize");
		for (const auto &ml : multilins) multilins_.push_back(Multilin{true, FSliceMut{const_cast<void *>(ml.ptr), ml.len_}});
		const size_t m = multilins.size();
		for (const auto &c : compositions) {
			ArithCircuit prod_expr = c.expression();
			prod_expr *= ArithCircuit::var(m); // add eq_ind (:399-407)
			evaluators_.push_back(hal.compile_expr(prod_expr));
		}
		state_ = InitialSums;
		sums_or_coeffs_ = sums;
		eq_ind_ = Multilin{true, FSliceMut{const_cast<void *>(eq_ind_partial_evals.ptr), eq_ind_partial_evals.len_}};
	}
	static size_t required_host_memory(size_t n_multilinears) { return n_multilinears + 1; }
	static size_t required_device_memory(size_t n_multilinears, size_t n_vars, bool with_eq_ind_partial_evals)
	{
		return (n_multilinears + (with_eq_ind_partial_evals ? 0 : 1)) * ((size_t)1 << (n_vars - 1));
	}
	size_t n_vars() const { return n_vars_initial_; }

	// round polynomial of degree 3 (:273-318)
	std::vector<B128> execute(B128 batch_coeff)
	{
		std::vector<FSlice> mls;
		for (const auto &m : multilins_) mls.push_back(FSlice{m.evals.ptr, m.evals.len_});
		const FSlice eq{eq_ind_.evals.ptr, eq_ind_.evals.len_};
		const std::vector<B128> round_evals = calculate_round_evals(hal_, n_vars_remaining_, batch_coeff, mls, evaluators_, &eq);
		B128 batched_sum;
		switch (state_) {
		case Coeffs: throw SumcheckError("ExpectedFold");
		case InitialSums: batched_sum = evaluate_univariate(sums_or_coeffs_, batch_coeff); break;
		default: batched_sum = batched_sum_; break;
		}
		const B128 alpha = eq_ind_challenges_[n_vars_remaining_ - 1];
		// calculate_round_coeffs_from_evals (:375-389)
		const B128 y_1 = round_evals[0], y_inf = round_evals[1];
		const B128 y_0 = (batched_sum - y_1 * alpha) * (B128::ONE() - alpha).invert_or_zero();
		const B128 c_0 = y_0, c_2 = y_inf, c_1 = y_1 - c_0 - c_2;
		const std::vector<B128> prime{c_0, c_1, c_2};
		state_ = Coeffs;
		sums_or_coeffs_ = prime;
		// v' -> v: eq(X, alpha) = (1 - alpha) + (2 alpha - 1) X   (:303-313)
		const B128 k0 = B128::ONE() - alpha, k1 = alpha.dbl() - B128::ONE();
		std::vector<B128> coeffs(4, B128::ZERO());
		for (size_t d = 0; d < 3; d++) {
			coeffs[d] = coeffs[d] + prime[d] * k0;
			coeffs[d + 1] = coeffs[d + 1] + prime[d] * k1;
		}
		for (auto &c : coeffs) c = c * eq_ind_prefix_eval_;
		return coeffs;
	}

	void fold(B128 challenge)
	{
		if (n_vars_remaining_ == 0) throw SumcheckError("ExpectedFinish");
		if (state_ != Coeffs) throw SumcheckError("ExpectedExecution");
		batched_sum_ = evaluate_univariate(sums_or_coeffs_, challenge);
		state_ = BatchedSum;
		eq_ind_prefix_eval_ = eq_ind_prefix_eval_ * eq(eq_ind_challenges_[n_vars_remaining_ - 1], challenge); // (:120-123)
		fold_multilinears(challenge);
		if (n_vars_remaining_ - 1 != 0) fold_eq_ind();
		n_vars_remaining_ -= 1;
	}

	// final evaluations followed by eq_ind_prefix_eval (:348-372)
	std::vector<B128> finish()
	{
		if (state_ == Coeffs) throw SumcheckError("ExpectedFold");
		if (n_vars_remaining_ != 0) throw SumcheckError("ExpectedExecution");
		HostSliceMut buffer = host_alloc_.alloc(multilins_.size());
		for (size_t i = 0; i < multilins_.size(); i++)
			hal_.copy_d2h(FSlice{multilins_[i].evals.ptr, multilins_[i].evals.len_}, &buffer[i], 1);
		std::vector<B128> res(buffer.ptr, buffer.ptr + multilins_.size());
		res.push_back(eq_ind_prefix_eval_);
		return res;
	}

private:
	struct Multilin {
		bool pre_fold;
		FSliceMut evals;
	};
	struct FoldArgs {
		FSliceMut evals_0;
		FSlice evals_1;
	};

	void fold_multilinears(B128 challenge) // (:145-193)
	{
		std::vector<FoldArgs> prepared;
		for (auto &m : multilins_) {
			if (m.pre_fold) {
				auto halves = ComputeMemory::split_half(FSlice{m.evals.ptr, m.evals.len_});
				FSliceMut folded = dev_alloc_.alloc((size_t)1 << (n_vars_remaining_ - 1));
				hal_.copy_d2d(halves.first, folded);
				prepared.push_back(FoldArgs{folded, halves.second});
			} else {
				auto halves = ComputeMemory::split_half_mut(m.evals);
				prepared.push_back(FoldArgs{halves.first, ComputeMemory::to_const(halves.second)});
			}
		}
		hal_.execute([&](ComputeLayerExecutor &exec) {
			multilins_ = exec.map(prepared.begin(), prepared.end(), [&](ComputeLayerExecutor &e, FoldArgs &a) {
				e.extrapolate_line(a.evals_0, a.evals_1, challenge);
				return Multilin{false, a.evals_0};
			});
			return std::vector<B128>{};
		});
	}

	void fold_eq_ind() // (:195-254): map_kernels { add_assign(evals_1 -> evals_0) }
	{
		const size_t split_n_vars = n_vars_remaining_ - 2;
		FSliceMut evals_0;
		FSlice evals_1;
		if (eq_ind_.pre_fold) {
			auto halves = ComputeMemory::split_half(FSlice{eq_ind_.evals.ptr, eq_ind_.evals.len_});
			evals_0 = dev_alloc_.alloc(halves.first.len());
			hal_.copy_d2d(halves.first, evals_0);
			evals_1 = halves.second;
		} else {
			auto halves = ComputeMemory::split_half_mut(eq_ind_.evals);
			evals_0 = halves.first;
			evals_1 = ComputeMemory::to_const(halves.second);
		}
		std::vector<KernelMemMap> kernel_mappings{KernelMemMap::chunked_mut(evals_0, 0), KernelMemMap::chunked(evals_1, 0)};
		hal_.execute([&](ComputeLayerExecutor &exec) {
			exec.map_kernels(
			    [&](KernelExecutor &local_exec, size_t log_chunks, std::vector<KernelBuffer> &buffers) {
				    const size_t log_chunk_size = split_n_vars - log_chunks;
				    local_exec.add_assign(log_chunk_size, buffers[1].to_ref(), buffers[0].as_mut());
			    },
			    kernel_mappings);
			return std::vector<B128>{};
		});
		eq_ind_ = Multilin{false, evals_0};
	}

	enum State { Coeffs, InitialSums, BatchedSum };
	ComputeLayer &hal_;
	DeviceBumpAllocator &dev_alloc_;
	HostBumpAllocator &host_alloc_;
	size_t n_vars_initial_, n_vars_remaining_;
	std::vector<Multilin> multilins_;
	std::vector<ExprEval> evaluators_;
	State state_;
	std::vector<B128> sums_or_coeffs_;
	B128 batched_sum_;
	B128 eq_ind_prefix_eval_ = B128::ONE();
	Multilin eq_ind_{};
	std::vector<B128> eq_ind_challenges_;
};

// ------------------------------------------------------------------------------------------------------------------
// BivariateMLEcheckProver, same statement and same transcript, with the equality indicator carried INSIDE one factor
// of every composition instead of being multiplied in at every hypercube point of every round.
//
// The reference's round r sums  eq_r(x') * a_r(X, x') * b_r(X, x')  over x' (bivariate_mlecheck.rs:391-520): three
// multilinears, two chained GF(2^128) products per point, and the 2^(n-1-r)-entry indicator table is folded next to
// the multilinears (:195-254).  Here one factor of each composition is replaced ONCE by
//     S_0(v, x') = a(v, x') * eq_0(x')                       (one element-wise product pass, compute_composite)
// and from then on the round is the plain bivariate product round of S and b -- the kernel the sumcheck prover runs,
// matrix-core Gram products and the fused fold + evaluation pass included.  What keeps S consistent:
//     eq_r(u, x'') = eq_{r+1}(x'') * (u ? zeta : 1 - zeta)    zeta = eq_ind_challenges[index of u]
// so after the fold with the round challenge, S_{r+1}(u, x'') = lambda * a_{r+1}(u, x'') * eq_{r+1}(x'') holds with
// ONE scalar lambda for both halves once the upper half (u = 1) is multiplied by (1 - zeta) / zeta:
// extrapolate_line_scaled.  lambda grows by (1 - zeta) per round and is divided out of the two round evaluations and
// of the final evaluation on the host.  Exact field arithmetic: the transcript is the reference's, bit for bit.
//
// Applicable when the compositions' graph has a proper 2-colouring (every product has exactly one weighted factor)
// and no indicator coordinate is 0 or 1 (zeta and 1 - zeta are inverted); the caller falls back to
// BivariateMLEcheckProver otherwise.  Device memory: 2^n per weighted multilinear, 2^(n-1) per other one.
class WeightedMLEcheckProver {
public:
	// colouring[i] = true: multilinear i carries the indicator.  Empty result = no proper colouring.
	static std::vector<bool> colouring(size_t n_multilinears, const std::vector<IndexCompositionBivariate> &compositions)
	{
		std::vector<int> col(n_multilinears, -1);
		std::vector<bool> none;
		// components in index order; the first vertex of a component is left unweighted unless that puts more
		// arrays on the weighted side than the other choice (weighted arrays cost twice the memory)
		for (size_t root = 0; root < n_multilinears; root++) {
			if (col[root] >= 0) continue;
			std::vector<size_t> comp{root};
			col[root] = 0;
			bool touched = false;
			for (size_t h = 0; h < comp.size(); h++)
				for (const auto &c : compositions)
					for (int side = 0; side < 2; side++)
						if (c.indices[side] == comp[h]) {
							touched = true;
							const size_t o = c.indices[1 - side];
							if (o >= n_multilinears) return none;
							if (col[o] < 0) {
								col[o] = 1 - col[comp[h]];
								comp.push_back(o);
							} else if (col[o] == col[comp[h]]) {
								return none; // odd cycle (or a square a * a)
							}
						}
			if (!touched) continue; // not in any composition: folded like an unweighted one
			size_t ones = 0;
			for (size_t v : comp) ones += col[v] == 1;
			if (2 * ones > comp.size())
				for (size_t v : comp) col[v] = 1 - col[v];
			// a component with a composition has at least one vertex of each colour
		}
		std::vector<bool> out(n_multilinears);
		for (size_t i = 0; i < n_multilinears; i++) out[i] = col[i] == 1;
		return out;
	}
	static bool coordinates_invertible(const std::vector<B128> &eq_ind_challenges, size_t n_vars)
	{
		for (size_t i = 0; i + 1 < n_vars; i++)
			if (eq_ind_challenges[i] == B128::ZERO() || eq_ind_challenges[i] == B128::ONE()) return false;
		return true;
	}
	static size_t required_device_memory(const std::vector<bool> &weighted, size_t n_vars)
	{
		size_t total = 0;
		for (bool w : weighted) total += w ? (size_t)1 << n_vars : (size_t)1 << (n_vars ? n_vars - 1 : 0);
		return total;
	}

	WeightedMLEcheckProver(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, HostBumpAllocator &host_alloc, size_t n_vars,
	                       const std::vector<IndexCompositionBivariate> &compositions, const std::vector<B128> &sums,
	                       const std::vector<FSlice> &multilins, FSlice eq_ind_partial_evals, std::vector<B128> eq_ind_challenges,
	                       std::vector<bool> weighted)
	    : hal_(hal), dev_alloc_(dev_alloc), host_alloc_(host_alloc), n_vars_initial_(n_vars), n_vars_remaining_(n_vars),
	      eq_ind_challenges_(std::move(eq_ind_challenges)), weighted_(std::move(weighted))
	{
		for (const auto &ml : multilins)
			if (ml.len() != (size_t)1 << n_vars) throw SumcheckError("NumberOfVariablesMismatch");
		if (eq_ind_partial_evals.len() != (size_t)1 << (n_vars ? n_vars - 1 : 0)) throw SumcheckError("IncorrectEqIndPartialEvalsSize");
		if (weighted_.size() != multilins.size()) throw SumcheckError("colouring does not match the multilinears");
		for (const auto &c : compositions) {
			if (weighted_[c.indices[0]] == weighted_[c.indices[1]]) throw SumcheckError("composition without exactly one weighted factor");
			evaluators_.push_back(hal.compile_expr(c.expression()));
		}
		state_ = InitialSums;
		sums_or_coeffs_ = sums;
		// S_0 = a * eq_0 on both halves of every weighted multilinear
		const ExprEval prod = hal.compile_expr(ArithCircuit::var(0) * ArithCircuit::var(1));
		const size_t half = eq_ind_partial_evals.len();
		for (size_t i = 0; i < multilins.size(); i++) {
			if (!weighted_[i]) {
				multilins_.push_back(Multilin{true, FSliceMut{const_cast<void *>(multilins[i].ptr), multilins[i].len_}});
				continue;
			}
			if (n_vars == 0) { // a single value, no indicator variables: nothing to weight
				multilins_.push_back(Multilin{true, FSliceMut{const_cast<void *>(multilins[i].ptr), multilins[i].len_}});
				continue;
			}
			FSliceMut s = dev_alloc_.alloc(multilins[i].len());
			auto in = ComputeMemory::split_half(multilins[i]);
			auto out = ComputeMemory::split_half_mut(s);
			hal.execute([&](ComputeLayerExecutor &exec) {
				exec.compute_composite(SlicesBatch<FSlice>({in.first, eq_ind_partial_evals}, half), out.first, prod);
				exec.compute_composite(SlicesBatch<FSlice>({in.second, eq_ind_partial_evals}, half), out.second, prod);
				return std::vector<B128>{};
			});
			multilins_.push_back(Multilin{false, s}); // our own buffer: folded in place from the first round on
		}
	}
	size_t n_vars() const { return n_vars_initial_; }

	std::vector<B128> execute(B128 batch_coeff)
	{
		std::vector<FSlice> mls;
		for (const auto &m : multilins_) mls.push_back(FSlice{m.evals.ptr, m.evals.len_});
		std::vector<B128> round_evals = calculate_round_evals(hal_, n_vars_remaining_, batch_coeff, mls, evaluators_);
		const B128 lambda_inv = lambda_.invert_or_zero();
		for (auto &e : round_evals) e = e * lambda_inv;
		B128 batched_sum;
		switch (state_) {
		case Coeffs: throw SumcheckError("ExpectedFold");
		case InitialSums: batched_sum = evaluate_univariate(sums_or_coeffs_, batch_coeff); break;
		default: batched_sum = batched_sum_; break;
		}
		// from here on: BivariateMLEcheckProver::execute (:273-318) unchanged
		const B128 alpha = eq_ind_challenges_[n_vars_remaining_ - 1];
		const B128 y_1 = round_evals[0], y_inf = round_evals[1];
		const B128 y_0 = (batched_sum - y_1 * alpha) * (B128::ONE() - alpha).invert_or_zero();
		const B128 c_0 = y_0, c_2 = y_inf, c_1 = y_1 - c_0 - c_2;
		const std::vector<B128> prime{c_0, c_1, c_2};
		state_ = Coeffs;
		sums_or_coeffs_ = prime;
		const B128 k0 = B128::ONE() - alpha, k1 = alpha.dbl() - B128::ONE();
		std::vector<B128> coeffs(4, B128::ZERO());
		for (size_t d = 0; d < 3; d++) {
			coeffs[d] = coeffs[d] + prime[d] * k0;
			coeffs[d + 1] = coeffs[d + 1] + prime[d] * k1;
		}
		for (auto &c : coeffs) c = c * eq_ind_prefix_eval_;
		return coeffs;
	}

	void fold(B128 challenge)
	{
		if (n_vars_remaining_ == 0) throw SumcheckError("ExpectedFinish");
		if (state_ != Coeffs) throw SumcheckError("ExpectedExecution");
		batched_sum_ = evaluate_univariate(sums_or_coeffs_, challenge);
		state_ = BatchedSum;
		eq_ind_prefix_eval_ = eq_ind_prefix_eval_ * eq(eq_ind_challenges_[n_vars_remaining_ - 1], challenge);
		// the variable that splits the folded arrays becomes the next round variable: level its two halves
		const bool scale = n_vars_remaining_ >= 2;
		B128 hi_scale = B128::ONE();
		if (scale) {
			const B128 zeta = eq_ind_challenges_[n_vars_remaining_ - 2];
			hi_scale = (B128::ONE() - zeta) * zeta.invert_or_zero();
			lambda_ = lambda_ * (B128::ONE() - zeta);
		}
		std::vector<FoldArgs> prepared;
		for (size_t i = 0; i < multilins_.size(); i++) {
			auto &m = multilins_[i];
			if (m.pre_fold) {
				auto halves = ComputeMemory::split_half(FSlice{m.evals.ptr, m.evals.len_});
				FSliceMut folded = dev_alloc_.alloc((size_t)1 << (n_vars_remaining_ - 1));
				hal_.copy_d2d(halves.first, folded);
				prepared.push_back(FoldArgs{folded, halves.second, scale && weighted_[i]});
			} else {
				auto halves = ComputeMemory::split_half_mut(m.evals);
				prepared.push_back(FoldArgs{halves.first, ComputeMemory::to_const(halves.second), scale && weighted_[i]});
			}
		}
		hal_.execute([&](ComputeLayerExecutor &exec) {
			multilins_ = exec.map(prepared.begin(), prepared.end(), [&](ComputeLayerExecutor &e, FoldArgs &a) {
				if (a.scaled)
					e.extrapolate_line_scaled(a.evals_0, a.evals_1, challenge, hi_scale);
				else
					e.extrapolate_line(a.evals_0, a.evals_1, challenge);
				return Multilin{false, a.evals_0};
			});
			return std::vector<B128>{};
		});
		n_vars_remaining_ -= 1;
	}

	std::vector<B128> finish()
	{
		if (state_ == Coeffs) throw SumcheckError("ExpectedFold");
		if (n_vars_remaining_ != 0) throw SumcheckError("ExpectedExecution");
		HostSliceMut buffer = host_alloc_.alloc(multilins_.size());
		for (size_t i = 0; i < multilins_.size(); i++)
			hal_.copy_d2h(FSlice{multilins_[i].evals.ptr, multilins_[i].evals.len_}, &buffer[i], 1);
		std::vector<B128> res(buffer.ptr, buffer.ptr + multilins_.size());
		const B128 lambda_inv = lambda_.invert_or_zero();
		for (size_t i = 0; i < res.size(); i++)
			if (weighted_[i]) res[i] = res[i] * lambda_inv;
		res.push_back(eq_ind_prefix_eval_);
		return res;
	}

private:
	struct Multilin {
		bool pre_fold;
		FSliceMut evals;
	};
	struct FoldArgs {
		FSliceMut evals_0;
		FSlice evals_1;
		bool scaled;
	};
	enum State { Coeffs, InitialSums, BatchedSum };
	ComputeLayer &hal_;
	DeviceBumpAllocator &dev_alloc_;
	HostBumpAllocator &host_alloc_;
	size_t n_vars_initial_, n_vars_remaining_;
	std::vector<Multilin> multilins_;
	std::vector<ExprEval> evaluators_;
	State state_;
	std::vector<B128> sums_or_coeffs_;
	B128 batched_sum_;
	B128 eq_ind_prefix_eval_ = B128::ONE();
	B128 lambda_ = B128::ONE();
	std::vector<B128> eq_ind_challenges_;
	std::vector<bool> weighted_;
};

} // namespace binius_amd
