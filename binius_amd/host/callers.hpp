// binius_amd/host/callers.hpp -- C++ mirrors of two more callers of the ComputeLayer in the reference's
// prover, the way sumcheck.hpp mirrors the sumcheck provers:
//   ProductCircuitLayers::compute   crates/core/src/protocols/prodcheck/prove.rs:24-77
//   RingSwitchEqInd                 crates/core/src/ring_switch/eq_ind.rs:39-141
#pragma once
#include "sumcheck.hpp"

namespace binius_amd {

// prodcheck/prove.rs:24-77: layer i of the binary product circuit (2^(i+1) values) is kept, each layer is the
// element-wise product of the halves of the layer below (compute_composite with the bivariate product)
class ProductCircuitLayers {
public:
	static ProductCircuitLayers compute(FSlice evals, ComputeLayer &hal, DeviceBumpAllocator &dev_alloc)
	{
		const size_t n = evals.len();
		if (n == 0 || (n & (n - 1)) != 0) throw Error(Error::InputValidation, "ExpectInputSlicePowerOfTwoLength");
		size_t log_n = 0;
		while (((size_t)1 << log_n) < n) log_n++;
		const ExprEval prod_expr = hal.compile_expr(ArithCircuit::var(0) * ArithCircuit::var(1)); // (:40-41)
		ProductCircuitLayers out;
		FSlice last_layer = evals;
		for (size_t i = log_n; i-- > 0;) {
			auto halves = ComputeMemory::split_half(last_layer);
			FSliceMut new_layer = dev_alloc.alloc((size_t)1 << i);
			hal.execute([&](ComputeLayerExecutor &exec) {
				exec.compute_composite(SlicesBatch<FSlice>({halves.first, halves.second}, (size_t)1 << i), new_layer, prod_expr);
				return std::vector<B128>{};
			});
			out.layers_.push_back(last_layer);
			last_layer = ComputeMemory::as_const(new_layer);
		}
		std::vector<B128> top(1);
		hal.copy_d2h(last_layer, top);
		out.product_ = top[0];
		std::reverse(out.layers_.begin(), out.layers_.end());
		return out;
	}
	const std::vector<FSlice> &layers() const { return layers_; }
	B128 product() const { return product_; }

private:
	std::vector<FSlice> layers_;
	B128 product_;
};

// ring_switch/eq_ind.rs:39-141.  kappa = log2 of the extension degree of F over the packed subfield
class RingSwitchEqInd {
public:
	struct Precompute { // RingSwitchEqIndPrecompute (:44-48)
		FSliceMut evals;
		FSlice row_batching_query_expansion;
		FSliceMut mle;
	};
	RingSwitchEqInd(std::vector<B128> z_vals, std::vector<B128> row_batch_coeffs, B128 mixing_coeff, size_t kappa)
	    : z_vals_(std::move(z_vals)), coeffs_(std::move(row_batch_coeffs)), mixing_coeff_(mixing_coeff), kappa_(kappa)
	{
		if (coeffs_.size() < ((size_t)1 << kappa)) // (:63-69)
			throw Error(Error::InputValidation,
			            "InvalidArgs(RingSwitchEqInd::new expects row_batch_coeffs length greater than or equal to the extension degree)");
	}
	// precompute_values (:78-121)
	Precompute precompute_values(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc) const
	{
		const size_t deg = (size_t)1 << kappa_;
		FSliceMut expansion = dev_alloc.alloc(deg);
		hal.copy_h2d(coeffs_.data(), deg, expansion);
		FSliceMut evals = dev_alloc.alloc((size_t)1 << z_vals_.size());
		hal.fill(evals, B128::ZERO()); // (the reference's allocators hand out zeroed memory for the part tensor_expand grows into)
		FSliceMut first = ComputeMemory::slice_power_of_two_mut(evals, 1);
		hal.fill(first, mixing_coeff_);
		FSliceMut mle = dev_alloc.alloc(evals.len());
		return Precompute{evals, ComputeMemory::as_const(expansion), mle};
	}
	// multilinear_extension (:123-141): tensor_expand(0, z_vals), then fold_right over the subfield limbs
	FSlice multilinear_extension(Precompute pre, ComputeLayerExecutor &exec) const
	{
		exec.tensor_expand(0, z_vals_, pre.evals);
		exec.fold_right(SubfieldSlice(ComputeMemory::as_const(pre.evals), 7 - kappa_), pre.row_batching_query_expansion, pre.mle);
		return ComputeMemory::as_const(pre.mle);
	}

private:
	std::vector<B128> z_vals_, coeffs_;
	B128 mixing_coeff_;
	size_t kappa_;
};

} // namespace binius_amd
