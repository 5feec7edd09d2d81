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

// byte B of w, masked with m (= 0xF0): the table offset (nibble * 16) of the HIGH nibble of that byte
// in one SDWA instruction (no shift).
template <int B>
__device__ __forceinline__ uint32_t byte_and(uint32_t w, uint32_t m)
{
	uint32_t r;
	if constexpr (B == 0)
		asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(m));
	else if constexpr (B == 1)
		asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(m));
	else if constexpr (B == 2)
		asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(m));
	else
		asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(m));
	return r;
}

__device__ __forceinline__ uint32_t ct_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// x * z via 32 conflict-free ds_read_b128.  ~105 VALU per element: per 32-bit word one rotate by 4
// (its low nibbles become the high nibbles of the rotated word's bytes), eight SDWA byte-selects
// that produce nibble*16 directly, and the 32 looked-up entries folded two at a time with the
// three-input XOR (v_bitop3_b32).
__device__ __forceinline__ uint4 ctable_mul(const ctable_smem &s, uint4 x)
{
	const char *base = reinterpret_cast<const char *>(s.T);
	uint4 acc{0, 0, 0, 0};
	const uint32_t w[4] = {x.x, x.y, x.z, x.w};
	const uint32_t m = 0xF0u;
#pragma unroll
	for (int wi = 0; wi < 4; wi++) {
		const uint32_t hi = w[wi];
		const uint32_t lo = __builtin_amdgcn_alignbit(hi, hi, 28); // rotl(w, 4)
		uint32_t off[8];
		off[0] = byte_and<0>(lo, m);
		off[1] = byte_and<0>(hi, m);
		off[2] = byte_and<1>(lo, m);
		off[3] = byte_and<1>(hi, m);
		off[4] = byte_and<2>(lo, m);
		off[5] = byte_and<2>(hi, m);
		off[6] = byte_and<3>(lo, m);
		off[7] = byte_and<3>(hi, m);
#pragma unroll
		for (int j = 0; j < 8; j += 2) {
			// byte offset of entry: table (8*wi + j) * 256 + nibble * 16
			const uint4 t0 = *reinterpret_cast<const uint4 *>(base + (8 * wi + j) * 256 + off[j]);
			const uint4 t1 = *reinterpret_cast<const uint4 *>(base + (8 * wi + j + 1) * 256 + off[j + 1]);
			acc.x = ct_xor3(acc.x, t0.x, t1.x);
			acc.y = ct_xor3(acc.y, t0.y, t1.y);
			acc.z = ct_xor3(acc.z, t0.z, t1.z);
			acc.w = ct_xor3(acc.w, t0.w, t1.w);
		}
	}
	return acc;
}

} // namespace bn
