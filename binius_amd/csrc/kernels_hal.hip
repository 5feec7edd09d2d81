// binius_amd/csrc/kernels_hal.hip -- the old HAL (binius_hal::ComputationBackend, crates/hal/src/backend.rs:35-84)
// on device-resident multilinears: the general forms of the sumcheck round calculation
// (crates/hal/src/sumcheck_round_calculation.rs:45-330) and of the single-variable fold with a constant suffix
// (crates/math/src/fold.rs:528-576, 648-696; crates/hal/src/sumcheck_folding.rs:114-143, 218-232).
//
// These are the GENERAL paths: any ArithCircuit composition, any set of evaluation points (0, 1, infinity,
// interpolation-domain points), both evaluation orders, truncated multilinears.  The shapes the provers actually
// spend their time in -- products of two multilinears evaluated at 1 and infinity, High-to-Low, with or without an
// equality-indicator factor -- are routed by abi_hal.cpp to the same kernels the ComputeLayer path uses
// (matrix-core Gram kernels / 9-lane kernels); what lands here is interpreted per hypercube point.
#include <hip/hip_runtime.h>

#include "ctable.hpp"
#include "internal.hpp"

namespace bn {

namespace {
constexpr int kHalMaxSteps = 64;

__device__ __forceinline__ f128 hal_circuit_eval(const bn_step *steps, uint32_t n_steps, const f128 *rows)
{
	f128 ev[kHalMaxSteps];
	for (uint32_t s = 0; s < n_steps; s++) {
		const bn_step st = steps[s];
		f128 r;
		switch (st.kind) {
		case BN_STEP_ADD: r = ev[st.a] ^ ev[st.b]; break;
		case BN_STEP_MUL: r = mul_slow(ev[st.a], ev[st.b]); break;
		case BN_STEP_POW: r = pow_slow(ev[st.a], st.b); break;
		case BN_STEP_CONST: r = f128{st.cst.lo, st.cst.hi}; break;
		default: r = rows[st.a]; break;
		}
		ev[s] = r;
	}
	return n_steps ? ev[n_steps - 1] : f128_zero();
}

__device__ __forceinline__ uint32_t hal_wave_xor(uint32_t v)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	return v;
}
} // namespace

// One evaluation point per pass over the cube: rows at the point, every evaluator that covers it, XOR-reduce.
__global__ __launch_bounds__(256) void k_hal_round_evals(hal_round_args a, f128 *out)
{
	__shared__ uint64_t red[4][2];
	const uint64_t half = (uint64_t)1 << (a.n_vars - 1);
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t p = a.pt_lo; p < a.pt_hi; p++) {
		f128 acc[kHalMaxEv];
		for (uint32_t e = 0; e < a.n_ev; e++)
			acc[e] = f128_zero();
		for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < half; i += (uint64_t)gridDim.x * 256) {
			f128 row[kHalMaxMl];
			const uint64_t i0 = a.order == BN_ORDER_LOW_TO_HIGH ? 2 * i : i;
			const uint64_t i1 = a.order == BN_ORDER_LOW_TO_HIGH ? 2 * i + 1 : i + half;
			for (uint32_t k = 0; k < a.n_ml; k++) {
				// stored value or the constant suffix (sumcheck_round_calculation.rs:421-441, 521-556)
				const f128 e0 = i0 < a.ml[k].len ? to_f128(a.ml[k].evals[i0]) : a.ml[k].suffix;
				const f128 e1 = i1 < a.ml[k].len ? to_f128(a.ml[k].evals[i1]) : a.ml[k].suffix;
				// f(z) = f(0) + z (f(1) - f(0)); index 2 = infinity: f(1) - f(0)   (:186-232)
				if (p == 0) row[k] = e0;
				else if (p == 1) row[k] = e1;
				else if (p == 2) row[k] = e0 ^ e1;
				else row[k] = e0 ^ mul_slow(a.pts[p - 3], e0 ^ e1);
			}
			for (uint32_t e = 0; e < a.n_ev; e++) {
				if (p < a.ev[e].pt_start || p >= a.ev[e].pt_end) continue;
				f128 v = p == 2 ? hal_circuit_eval(a.ev[e].steps_inf, a.ev[e].n_steps_inf, row) : hal_circuit_eval(a.ev[e].steps, a.ev[e].n_steps, row);
				if (a.ev[e].eq) v = mul_slow(v, to_f128(a.ev[e].eq[i]));
				acc[e] ^= v;
			}
		}
		for (uint32_t e = 0; e < a.n_ev; e++) {
			if (p < a.ev[e].pt_start || p >= a.ev[e].pt_end) continue;
			uint32_t w[4] = {(uint32_t)acc[e].lo, (uint32_t)(acc[e].lo >> 32), (uint32_t)acc[e].hi, (uint32_t)(acc[e].hi >> 32)};
#pragma unroll
			for (int t = 0; t < 4; t++)
				w[t] = hal_wave_xor(w[t]);
			__syncthreads();
			if (lane == 0) {
				red[wave][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
				red[wave][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
			}
			__syncthreads();
			if (threadIdx.x < 2) {
				const uint64_t v = red[0][threadIdx.x] ^ red[1][threadIdx.x] ^ red[2][threadIdx.x] ^ red[3][threadIdx.x];
				if (v)
					atomicXor(reinterpret_cast<unsigned long long *>(out + a.ev[e].out_off + (p - a.ev[e].pt_start)) + threadIdx.x,
					          (unsigned long long)v);
			}
		}
	}
}

hipError_t launch_hal_round_evals(hipStream_t s, int n_cu, const hal_round_args &a, f128 *d_out)
{
	const uint64_t half = (uint64_t)1 << (a.n_vars - 1);
	uint64_t blocks = (half + 255) / 256;
	const uint64_t cap = (uint64_t)n_cu * 4;
	hipLaunchKernelGGL(k_hal_round_evals, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, s, a, d_out);
	return hipGetLastError();
}

// out[i] = x0 + z (x1 - x0) with (x0, x1) = entries (2i, 2i+1) [Low-to-High] or (i, i + half) [High-to-Low] of a
// multilinear that stores `len` evaluations and is `suffix` beyond them; n_out outputs.  The constant
// multiplication goes through the LDS nibble tables (ctable.hpp).
__global__ __launch_bounds__(256) void k_hal_fold_lerp(const uint4 *evals, uint64_t len, f128 suffix, uint32_t order, uint64_t half, f128 z,
                                                       uint4 *out, uint64_t n_out)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	const uint4 sfx = to_u4(suffix);
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (uint64_t)gridDim.x * 256) {
		const uint64_t i0 = order == BN_ORDER_LOW_TO_HIGH ? 2 * i : i;
		const uint64_t i1 = order == BN_ORDER_LOW_TO_HIGH ? 2 * i + 1 : i + half;
		const uint4 x0 = i0 < len ? evals[i0] : sfx;
		const uint4 x1 = i1 < len ? evals[i1] : sfx;
		out[i] = xor4(x0, ctable_mul_pinned<8>(tab, xor4(x0, x1)));
	}
}

// One row of the general round calculation: the multilinear at evaluation point 0 / 1 / infinity (no multiplication: a
// gather, or one XOR) -- what brings Low-to-High pairs and truncated multilinears into the layout the throughput kernels read.
__global__ __launch_bounds__(256) void k_hal_row(const uint4 *evals, uint64_t len, f128 suffix, uint32_t order, uint64_t half, uint32_t point, uint4 *out,
                                                  uint64_t n_out)
{
	const uint4 sfx = to_u4(suffix);
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (uint64_t)gridDim.x * 256) {
		const uint64_t i0 = order == BN_ORDER_LOW_TO_HIGH ? 2 * i : i;
		const uint64_t i1 = order == BN_ORDER_LOW_TO_HIGH ? 2 * i + 1 : i + half;
		uint4 r;
		if (point == 0) {
			r = i0 < len ? evals[i0] : sfx;
		} else if (point == 1) {
			r = i1 < len ? evals[i1] : sfx;
		} else {
			const uint4 x0 = i0 < len ? evals[i0] : sfx;
			const uint4 x1 = i1 < len ? evals[i1] : sfx;
			r = xor4(x0, x1);
		}
		out[i] = r;
	}
}

// All the rows of a general round calculation in ONE launch (blockIdx.y = job): at 2^19 points a row is a few microseconds
// of kernel behind as many of launch, and a request with three multilinears and three points has six of them.
__global__ __launch_bounds__(256) void k_hal_rows(hal_rows_args a)
{
	__shared__ ctable_smem tab;
	const hal_rows_args::job j = a.jobs[blockIdx.y];
	if (j.point > 2) ctable_build(tab, j.z); // (uniform per workgroup)
	const uint4 sfx = to_u4(j.suffix);
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_out; i += (uint64_t)gridDim.x * 256) {
		const uint64_t i0 = a.order == BN_ORDER_LOW_TO_HIGH ? 2 * i : i;
		const uint64_t i1 = a.order == BN_ORDER_LOW_TO_HIGH ? 2 * i + 1 : i + a.half;
		uint4 r;
		if (j.point == 0) {
			r = i0 < j.len ? j.evals[i0] : sfx;
		} else if (j.point == 1) {
			r = i1 < j.len ? j.evals[i1] : sfx;
		} else {
			const uint4 x0 = i0 < j.len ? j.evals[i0] : sfx;
			const uint4 x1 = i1 < j.len ? j.evals[i1] : sfx;
			r = xor4(x0, x1);
			if (j.point > 2) r = xor4(x0, ctable_mul_pinned<8>(tab, r));
		}
		j.out[i] = r;
	}
}

hipError_t launch_hal_rows(hipStream_t s, int n_cu, const hal_rows_args &a)
{
	if (a.n_jobs == 0 || a.n_out == 0) return hipSuccess;
	uint64_t blocks = (a.n_out + 255) / 256;
	const uint64_t cap = ((uint64_t)n_cu * 8 + a.n_jobs - 1) / a.n_jobs;
	if (blocks > cap) blocks = cap ? cap : 1;
	hipLaunchKernelGGL(k_hal_rows, dim3((unsigned)blocks, a.n_jobs), dim3(256), 0, s, a);
	return hipGetLastError();
}

hipError_t launch_hal_row(hipStream_t s, int n_cu, const void *evals, uint64_t len, f128 suffix, uint32_t order, uint64_t half, uint32_t point, f128 z,
                          void *out, uint64_t n_out)
{
	if (n_out == 0) return hipSuccess;
	if (point > 2) return launch_hal_fold_lerp(s, n_cu, evals, len, suffix, order, half, z, out, n_out);
	uint64_t blocks = (n_out + 255) / 256;
	const uint64_t cap = (uint64_t)n_cu * 8;
	hipLaunchKernelGGL(k_hal_row, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, s, (const uint4 *)evals, len, suffix, order, half, point,
	                   (uint4 *)out, n_out);
	return hipGetLastError();
}

hipError_t launch_hal_fold_lerp(hipStream_t s, int n_cu, const void *evals, uint64_t len, f128 suffix, uint32_t order, uint64_t half, f128 z,
                                void *out, uint64_t n_out)
{
	if (n_out == 0) return hipSuccess;
	uint64_t blocks = (n_out + 255) / 256;
	const uint64_t cap = (uint64_t)n_cu * 8;
	hipLaunchKernelGGL(k_hal_fold_lerp, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, s, (const uint4 *)evals, len, suffix, order, half, z,
	                   (uint4 *)out, n_out);
	return hipGetLastError();
}

} // namespace bn
