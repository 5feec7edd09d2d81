// binius_amd/csrc/re9.hpp -- pieces shared by the 9-lane Karatsuba round-evaluation kernels
// (kernels_roundeval9.hip, kernels_foldeval9.hip): wave geometry, the per-wave LDS tile layout,
// the GF(2^32)->GF(2^128) recombination and the workgroup tail (collapse, atomic XOR, fused finalize).
#pragma once
#include <hip/hip_runtime.h>

#include "bitslice.hpp"
#include "finalize.hpp"
#include "internal.hpp"

namespace bn {
namespace re9 {

constexpr int kGroups = 7;          // 9-lane groups per wave
constexpr int kPts = 16;            // hypercube points per group per batch
constexpr int kBatch = kGroups * kPts; // 112 points per wave-batch
constexpr int kBlkQ = 9;            // LDS uint4 per (limb, group) block: 32 planes + 16 B pad (bank spread)
constexpr int kZeroBlk = 8 * kGroups; // block of zeros for unused combination slots
constexpr int kWaveQ = (kZeroBlk + 1) * kBlkQ;

__device__ __forceinline__ uint32_t wave_xor_u32(uint32_t v)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	return v;
}

// (z0, z2, z1') of a Karatsuba level -> the product, for 32-bit pieces: T_6 element from T_5 parts
// lo = z0 + z2 ; hi = z1' + z0 + z2 + z2 * X_4      (pairwise_recursive_arithmetic.rs:18-28)
__device__ __forceinline__ uint64_t combine32(uint32_t z0, uint32_t z2, uint32_t z1p)
{
	const uint32_t lo = z0 ^ z2;
	const uint32_t hi = z1p ^ lo ^ (uint32_t)mulx64<4>((uint64_t)z2);
	return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ f128 combine64(uint64_t Z0, uint64_t Z2, uint64_t Z1p)
{
	const uint64_t lo = Z0 ^ Z2;
	return f128{lo, Z1p ^ lo ^ mulx64<5>(Z2)};
}

// combination mask over limbs 0..3 owned by lane slot c of a group
__device__ __forceinline__ unsigned combo_mask(unsigned c)
{
	switch (c) {
	case 0: return 1;
	case 1: return 2;
	case 2: return 3;
	case 3: return 4;
	case 4: return 8;
	case 5: return 12;
	case 6: return 5;
	case 7: return 10;
	default: return 15;
	}
}

// Workgroup tail: every lane holds 32 accumulator planes (low 16 bits: first stream, high 16 bits:
// second stream).  Collapses them, recombines the 9 limb products of every wave into field elements,
// XORs the two sums into out[0], out[1] and, if asked, runs the fused finalize in the last workgroup.
// Must be called by all NW*64 threads of the workgroup.
template <int NW = 4>
__device__ __forceinline__ void tail(const uint32_t (&acc)[32], bool live, unsigned c, unsigned g, unsigned wave, unsigned lane,
                                     f128 *out, const fin_fuse &fz, uint64_t seq, const fin_cache *fc = nullptr)
{
	__shared__ uint32_t red[NW][2][9][8];
	__shared__ uint64_t wsum[NW][4];
	// ---- collapse: per lane two GF(2^32) partial sums (low half -> S_1 / first stream, high -> S_inf)
	uint32_t s_lo = 0, s_hi = 0;
#pragma unroll
	for (int p = 0; p < 32; p++) {
		s_lo |= (__popc(acc[p] & 0xFFFFu) & 1u) << p;
		s_hi |= (__popc(acc[p] >> 16) & 1u) << p;
	}
	if (live) {
		red[wave][0][c][g] = s_lo;
		red[wave][1][c][g] = s_hi;
	}
	__syncthreads();
	BN_TS(6);
	if (lane < 2) {
		// lane q of every wave recombines stream q of that wave
		uint32_t pc[9];
#pragma unroll
		for (int cc = 0; cc < 9; cc++) {
			uint32_t v = 0;
#pragma unroll
			for (int gg = 0; gg < kGroups; gg++)
				v ^= red[wave][lane][cc][gg];
			pc[cc] = v;
		}
		const uint64_t Z0 = combine32(pc[0], pc[1], pc[2]);
		const uint64_t Z2 = combine32(pc[3], pc[4], pc[5]);
		const uint64_t Z1p = combine32(pc[6], pc[7], pc[8]);
		const f128 S = combine64(Z0, Z2, Z1p);
		wsum[wave][2 * lane] = S.lo;
		wsum[wave][2 * lane + 1] = S.hi;
	}
	__syncthreads();
	BN_TS(7);
	// fc: the finalize arguments were staged in LDS at kernel entry (finalize.hpp) -- nothing below touches the
	// kernarg segment then
	unsigned *const counter = fc ? fc->counter : fz.counter;
	f128 *const S = fc ? fc->S : fz.S;
	if (counter && gridDim.x == 1 && out == S) {
		// single workgroup: the sums never leave the chip -- finalize straight from LDS
		__shared__ f128 s_loc[2];
		if (threadIdx.x < 4) {
			uint64_t v = 0;
#pragma unroll
			for (int ww = 0; ww < NW; ww++)
				v ^= wsum[ww][threadIdx.x];
			reinterpret_cast<uint64_t *>(s_loc)[threadIdx.x] = v;
		}
		__syncthreads();
		if (fc)
			finalize_cached(*fc, seq, s_loc);
		else
			finalize_body(fz.args, fz.S, fz.rets, fz.mail, seq, s_loc, &fz.peer);
		return;
	}
	if (threadIdx.x < 4) {
		uint64_t v = 0;
#pragma unroll
		for (int ww = 0; ww < NW; ww++)
			v ^= wsum[ww][threadIdx.x];
		if (v)
			atomicXor(reinterpret_cast<unsigned long long *>(out) + threadIdx.x, (unsigned long long)v);
	}
	if (counter) {
		// fused finalize: release our partial, take a ticket; the last workgroup folds the sums into
		// the kernel's values and publishes them (same body as k_finalize)
		// No release/acquire FENCES here (an agent-scope release writes back the whole L2: several
		// microseconds per workgroup, measured +14 us per launch): every shared word is touched only by
		// device-scope atomics, which are performed at the coherence point.  Each XOR-ing lane drains
		// its own atomic (vmcnt(0)), the barrier orders the lanes, then one lane takes the ticket.
		__shared__ unsigned is_last;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			const unsigned t = atomicAdd(counter, 1u);
			is_last = (t == gridDim.x - 1) ? 1u : 0u;
		}
		__syncthreads();
		if (is_last) {
			// S is read with agent-scope atomic loads inside finalize_body (they bypass the L1)
			if (fc)
				finalize_cached(*fc, seq);
			else
				finalize_body(fz.args, fz.S, fz.rets, fz.mail, seq, nullptr, &fz.peer);
			if (threadIdx.x == 0)
				__hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

} // namespace re9
} // namespace bn
