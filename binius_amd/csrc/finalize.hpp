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

// Must be called by EVERY thread of a workgroup with >= 128 threads (it contains barriers).
// S_local == nullptr: the raw sums are in global memory (S, XOR-accumulated by all workgroups) and are
// read with agent-scope atomic loads, all terms at once (one memory round trip), then re-zeroed.
// S_local != nullptr: a single-workgroup launch kept its sums in LDS (S_local[slot]); global S was
// never touched.
// Term t is folded in by 128 threads (lane i contributes bit_i(S) ? coeff * 2^i : 0, gf128.hpp
// mul_basis); a workgroup of 256 threads handles two terms per pass.
__device__ __forceinline__ void finalize_body(const fin_args &a, f128 *S, f128 *rets, f128 *mail, uint64_t seq,
                                              const f128 *S_local = nullptr)
{
	__shared__ f128 fin_S[kFinMaxTerms];
	__shared__ uint64_t fin_red[4][2][2];   // [term group][wave in group][lo/hi]
	__shared__ f128 fin_values[kFinMaxValues];
	const unsigned tid = threadIdx.x;
	const unsigned n_groups = blockDim.x >= 512 ? 4u : (blockDim.x >= 256 ? 2u : 1u);
	if (tid < a.n_values)
		fin_values[tid] = a.init[tid];
	if (tid < a.n_terms) {
		const uint32_t slot = a.terms[tid].slot;
		f128 s;
		if (S_local) {
			s = S_local[slot];
		} else {
			s.lo = __hip_atomic_load(&S[slot].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s.hi = __hip_atomic_load(&S[slot].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		fin_S[tid] = s;
	}
	__syncthreads();
	const unsigned grp = tid >> 7, l128 = tid & 127;
	for (uint32_t t0 = 0; t0 < a.n_terms; t0 += n_groups) {
		const uint32_t t = t0 + grp;
		f128 c = f128_zero();
		if (grp < n_groups && t < a.n_terms) {
			const f128 s = fin_S[t];
			const f128 coeff = a.terms[t].coeff;
			if (coeff.lo == 1 && coeff.hi == 0) {
				// batch coefficient alpha^0 = 1 (the only one on the measured single-claim path)
				if (l128 == 0) c = s;
			} else {
				const uint64_t word = l128 < 64 ? s.lo : s.hi;
				if ((word >> (l128 & 63)) & 1)
					c = mul_basis(coeff, l128);
			}
		}
		uint32_t w[4] = {(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32)};
#pragma unroll
		for (int q = 0; q < 4; q++)
			w[q] = fin_wave_xor(w[q]);
		if (grp < n_groups && (tid & 63) == 0) {
			fin_red[grp][l128 >> 6][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
			fin_red[grp][l128 >> 6][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
		}
		__syncthreads();
		if (tid == 0) {
			for (unsigned gq = 0; gq < n_groups && t0 + gq < a.n_terms; gq++) {
				const uint32_t v = a.terms[t0 + gq].value;
				fin_values[v].lo ^= fin_red[gq][0][0] ^ fin_red[gq][1][0];
				fin_values[v].hi ^= fin_red[gq][0][1] ^ fin_red[gq][1][1];
			}
		}
		__syncthreads();
	}
	if (tid < a.n_ret) {
		const f128 v = fin_values[a.ret_ids[tid]];
		rets[tid] = v;
		if (seq) {
			// zero-copy return: values, then the sequence word, into fine-grained host memory
			__hip_atomic_store(&mail[tid].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(&mail[tid].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	// leave the accumulator slots zero for the next launch (no memset on the per-round path)
	if (!S_local && tid < a.n_slots) {
		__hip_atomic_store(&S[tid].lo, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(&S[tid].hi, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (seq) {
		// n_ret <= 8: the value stores above were issued by lanes of wave 0; the release below makes
		// wave 0 drain them (vmcnt) and write them through before the sequence word
		if (tid == 0)
			__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

} // namespace bn
