// binius_amd/csrc/ctable.hpp -- multiplication by a launch-constant GF(2^128) element through
// nibble tables staged in LDS.
//
// x -> x*z is GF(2)-linear, so x*z = XOR_p T[p][nibble_p(x)] with T[p][e] = (e << 4p) * z.
// 32 tables x 16 entries x 16 B = 8 KiB.  One table is exactly one 256-byte LDS bank row, entry e
// covers banks 4e..4e+3: lanes that read the same entry broadcast, lanes that read different
// entries hit different banks, so every ds_read_b128 is conflict-free no matter what the data is
// (a byte-indexed 64 KiB table would be ~3x slower per lookup from bank conflicts).
//
// The table is built by the workgroup itself from z * 2^i = z * prod X_k (gf128.hpp): 128 lanes
// each apply at most seven SWAR mulx steps, then 512 entries are XORs of <= 4 basis products.
// No host pre-pass, no extra launch, no global table.
#pragma once
#include <hip/hip_runtime.h>

#include "gf128.hpp"

namespace bn {

struct ctable_smem {
	uint4 T[512];    // [p][e]
	uint4 basis[128]; // z * 2^i
};

__device__ __forceinline__ uint4 to_u4(f128 v)
{
	return uint4{(uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32)};
}
__device__ __forceinline__ f128 to_f128(uint4 v)
{
	return f128{(uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32)};
}
__device__ __forceinline__ uint4 xor4(uint4 a, uint4 b) { return uint4{a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }

// Requires blockDim.x >= 128.  Ends with __syncthreads().
__device__ __forceinline__ void ctable_build(ctable_smem &s, f128 z)
{
	const unsigned tid = threadIdx.x;
	if (tid < 128)
		s.basis[tid] = to_u4(mul_basis(z, tid));
	__syncthreads();
	for (unsigned e = tid; e < 512; e += blockDim.x) {
		const unsigned p4 = (e >> 4) << 2;
		uint4 v{0, 0, 0, 0};
		if (e & 1) v = xor4(v, s.basis[p4]);
		if (e & 2) v = xor4(v, s.basis[p4 + 1]);
		if (e & 4) v = xor4(v, s.basis[p4 + 2]);
		if (e & 8) v = xor4(v, s.basis[p4 + 3]);
		s.T[e] = v;
	}
	__syncthreads();
}

// x * z via 32 conflict-free ds_read_b128.
__device__ __forceinline__ uint4 ctable_mul(const ctable_smem &s, uint4 x)
{
	const char *base = reinterpret_cast<const char *>(s.T);
	uint4 acc{0, 0, 0, 0};
	const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
	for (int wi = 0; wi < 4; wi++) {
#pragma unroll
		for (int j = 0; j < 8; j++) {
			// byte offset of entry: table (8*wi + j) * 256 + nibble * 16
			uint32_t off = (j == 0) ? ((w[wi] << 4) & 0xF0u) : ((w[wi] >> (4 * j - 4)) & 0xF0u);
			const uint4 t = *reinterpret_cast<const uint4 *>(base + (8 * wi + j) * 256 + off);
			acc = xor4(acc, t);
		}
	}
	return acc;
}

} // namespace bn
