// binius_amd/csrc/finalize.hpp -- the tail of every accumulate_kernels launch: fold the raw device
// sums into the kernel's declared values (value = init + sum_t coeff_t * S[slot_t], the
// `*accumulator += ret * batch_coeff` of crates/compute/src/cpu/layer.rs:512), gather the returned
// values, re-zero the accumulator slots and publish to the host mailbox.
//
// Shared by the stand-alone k_finalize kernel and by the LAST workgroup of the round-eval kernel
// (fused form: one launch per round evaluation instead of two).
#pragma once
#include <hip/hip_runtime.h>

#include "gf128.hpp"
#include "internal.hpp"

namespace bn {

__device__ __forceinline__ uint32_t fin_wave_xor(uint32_t v)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	return v;
}

// Must be called by EVERY thread of a workgroup with >= 128 threads (it contains barriers); the
// first 128 threads do the work.  S is read with agent-scope atomic loads so the fused caller sees
// the other workgroups' atomicXor results.
__device__ __forceinline__ void finalize_body(const fin_args &a, f128 *S, f128 *rets, f128 *mail, uint64_t seq)
{
	__shared__ uint64_t fin_red[2][2];
	__shared__ f128 fin_values[kFinMaxValues];
	const unsigned tid = threadIdx.x;
	if (tid < a.n_values)
		fin_values[tid] = a.init[tid];
	__syncthreads();
	for (uint32_t t = 0; t < a.n_terms; t++) {
		const fin_term tm = a.terms[t];
		f128 s;
		s.lo = __hip_atomic_load(&S[tm.slot].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		s.hi = __hip_atomic_load(&S[tm.slot].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		f128 c = f128_zero();
		if (tid < 128) {
			if (tm.coeff.lo == 1 && tm.coeff.hi == 0) {
				// batch coefficient alpha^0 = 1 (the only one on the measured single-claim path)
				if (tid == 0) c = s;
			} else {
				// lane i contributes bit_i(S) ? coeff * 2^i : 0  (gf128.hpp mul_basis)
				const uint64_t word = tid < 64 ? s.lo : s.hi;
				if ((word >> (tid & 63)) & 1)
					c = mul_basis(tm.coeff, tid);
			}
		}
		uint32_t w[4] = {(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32)};
#pragma unroll
		for (int q = 0; q < 4; q++)
			w[q] = fin_wave_xor(w[q]);
		__syncthreads();
		if (tid < 128 && (tid & 63) == 0) {
			fin_red[tid >> 6][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
			fin_red[tid >> 6][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
		}
		__syncthreads();
		if (tid == 0) {
			fin_values[tm.value].lo ^= fin_red[0][0] ^ fin_red[1][0];
			fin_values[tm.value].hi ^= fin_red[0][1] ^ fin_red[1][1];
		}
	}
	__syncthreads();
	if (tid < a.n_ret)
		rets[tid] = fin_values[a.ret_ids[tid]];
	// leave the accumulator slots zero for the next launch (no memset on the per-round path)
	if (tid < a.n_slots) {
		__hip_atomic_store(&S[tid].lo, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(&S[tid].hi, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (seq) {
		// zero-copy return: values, then the sequence word, into fine-grained host memory
		if (tid < a.n_ret) {
			const f128 v = fin_values[a.ret_ids[tid]];
			__hip_atomic_store(&mail[tid].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(&mail[tid].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		__threadfence_system();
		__syncthreads();
		if (tid == 0)
			__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

} // namespace bn
