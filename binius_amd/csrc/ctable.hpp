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

// Group form: `nthreads` (>= 128) consecutive threads, numbered ltid, build table s for z; every
// thread of the workgroup must call it (it contains workgroup barriers), threads outside any group
// pass ltid >= nthreads.  Several groups can build different tables at the same time.
__device__ __forceinline__ void ctable_build_group(ctable_smem &s, f128 z, unsigned ltid, unsigned nthreads)
{
	if (ltid < 128)
		s.basis[ltid] = to_u4(mul_basis(z, ltid));
	__syncthreads();
	for (unsigned e = ltid; e < 512; e += nthreads) {
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
// The lookups are issued in groups of G (default 16 = two words) before anything consumes them: a
// dependent read-xor chain would keep only 2-4 reads in flight and turn the multiplication into
// 8-16 serialized LDS round trips (measured: the fused fold+eval kernel 222 -> see profiles/r01).
template <int G = 16>
__device__ __forceinline__ uint4 ctable_mul(const ctable_smem &s, uint4 x)
{
	static_assert(G == 8 || G == 16 || G == 32, "group = 1, 2 or 4 words of lookups in flight");
	const char *base = reinterpret_cast<const char *>(s.T);
	uint4 acc{0, 0, 0, 0};
	const uint32_t w[4] = {x.x, x.y, x.z, x.w};
	const uint32_t m = 0xF0u;
	constexpr int WPG = G / 8; // words per group
#pragma unroll
	for (int w0 = 0; w0 < 4; w0 += WPG) {
		uint32_t off[G];
#pragma unroll
		for (int wq = 0; wq < WPG; wq++) {
			const uint32_t hi = w[w0 + wq];
			const uint32_t lo = __builtin_amdgcn_alignbit(hi, hi, 28); // rotl(w, 4)
			off[8 * wq + 0] = byte_and<0>(lo, m);
			off[8 * wq + 1] = byte_and<0>(hi, m);
			off[8 * wq + 2] = byte_and<1>(lo, m);
			off[8 * wq + 3] = byte_and<1>(hi, m);
			off[8 * wq + 4] = byte_and<2>(lo, m);
			off[8 * wq + 5] = byte_and<2>(hi, m);
			off[8 * wq + 6] = byte_and<3>(lo, m);
			off[8 * wq + 7] = byte_and<3>(hi, m);
		}
		uint4 t[G];
#pragma unroll
		for (int j = 0; j < G; j++) // byte offset of entry: table (8*word + nibble_index) * 256 + nibble * 16
			t[j] = *reinterpret_cast<const uint4 *>(base + (8 * w0 + j) * 256 + off[j]);
		__builtin_amdgcn_sched_barrier(0); // keep the G reads ahead of their consumers
#pragma unroll
		for (int j = 0; j < G; j += 2) {
			acc.x = ct_xor3(acc.x, t[j].x, t[j + 1].x);
			acc.y = ct_xor3(acc.y, t[j].y, t[j + 1].y);
			acc.z = ct_xor3(acc.z, t[j].z, t[j + 1].z);
			acc.w = ct_xor3(acc.w, t[j].w, t[j + 1].w);
		}
	}
	return acc;
}

// Same product with the lookup schedule pinned by compiler barriers: groups of G lookups, group g + 1
// requested before group g is folded in, nothing else in flight.  ctable_mul leaves the placement of the
// reads to the scheduler, which (when the surrounding kernel leaves it head-room) hoists all 32 reads of
// an element -- and of the next elements -- to the front and spills hundreds of registers; the kernels
// that also hold MFMA accumulators cannot afford that.
template <int G = 4>
__device__ __forceinline__ uint4 ctable_mul_pinned(const ctable_smem &s, uint4 x)
{
	static_assert(G == 4 || G == 8, "half a word or one word of lookups per group");
	const char *base = reinterpret_cast<const char *>(s.T);
	const uint32_t w[4] = {x.x, x.y, x.z, x.w};
	const uint32_t m = 0xF0u;
	uint32_t off[32];
#pragma unroll
	for (int wq = 0; wq < 4; wq++) {
		const uint32_t hi = w[wq];
		const uint32_t lo = __builtin_amdgcn_alignbit(hi, hi, 28); // rotl(w, 4)
		off[8 * wq + 0] = byte_and<0>(lo, m);
		off[8 * wq + 1] = byte_and<0>(hi, m);
		off[8 * wq + 2] = byte_and<1>(lo, m);
		off[8 * wq + 3] = byte_and<1>(hi, m);
		off[8 * wq + 4] = byte_and<2>(lo, m);
		off[8 * wq + 5] = byte_and<2>(hi, m);
		off[8 * wq + 6] = byte_and<3>(lo, m);
		off[8 * wq + 7] = byte_and<3>(hi, m);
	}
	uint4 acc{0, 0, 0, 0};
	uint4 cur[G], nxt[G];
#pragma unroll
	for (int j = 0; j < G; j++)
		cur[j] = *reinterpret_cast<const uint4 *>(base + j * 256 + off[j]);
#pragma unroll
	for (int g0 = 0; g0 < 32; g0 += G) {
		if (g0 + G < 32) {
#pragma unroll
			for (int j = 0; j < G; j++)
				nxt[j] = *reinterpret_cast<const uint4 *>(base + (g0 + G + j) * 256 + off[g0 + G + j]);
		}
		asm volatile("" ::: "memory");
#pragma unroll
		for (int j = 0; j < G; j += 2) {
			acc.x = ct_xor3(acc.x, cur[j].x, cur[j + 1].x);
			acc.y = ct_xor3(acc.y, cur[j].y, cur[j + 1].y);
			acc.z = ct_xor3(acc.z, cur[j].z, cur[j + 1].z);
			acc.w = ct_xor3(acc.w, cur[j].w, cur[j + 1].w);
		}
		asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w)::"memory");
#pragma unroll
		for (int j = 0; j < G; j++)
			cur[j] = nxt[j];
	}
	return acc;
}

// Variant of ctable_mul_pinned for measurements (tools/gram_bench.hip, FE_VARIANT): returns init ^ x * z -- the fold's
// "x0 +" rides in the first three-input XOR instead of costing four more --, and with LAZY the table offsets of a group are
// formed right before its reads are issued (4 live offset registers instead of 32).
template <int G, bool LAZY>
__device__ __forceinline__ uint4 ctable_mul_acc(const ctable_smem &s, uint4 x, uint4 init)
{
	static_assert(G == 4 || G == 8, "half a word or one word of lookups per group");
	const char *base = reinterpret_cast<const char *>(s.T);
	const uint32_t w[4] = {x.x, x.y, x.z, x.w};
	const uint32_t m = 0xF0u;
	uint32_t off[32];
	uint32_t rot[4];
	auto offset = [&](int j) -> uint32_t { // j = 8 * word + r: r even -> low nibble of byte r / 2 (from the rotated word), odd -> its high nibble
		const int wq = j >> 3, r = j & 7;
		const uint32_t src = (r & 1) ? w[wq] : rot[wq];
		switch (r >> 1) {
		case 0: return byte_and<0>(src, m);
		case 1: return byte_and<1>(src, m);
		case 2: return byte_and<2>(src, m);
		default: return byte_and<3>(src, m);
		}
	};
#pragma unroll
	for (int wq = 0; wq < 4; wq++)
		rot[wq] = __builtin_amdgcn_alignbit(w[wq], w[wq], 28); // rotl(w, 4)
	if constexpr (!LAZY) {
#pragma unroll
		for (int j = 0; j < 32; j++)
			off[j] = offset(j);
	}
	uint4 acc = init;
	uint4 cur[G], nxt[G];
#pragma unroll
	for (int j = 0; j < G; j++)
		cur[j] = *reinterpret_cast<const uint4 *>(base + j * 256 + (LAZY ? offset(j) : off[j]));
#pragma unroll
	for (int g0 = 0; g0 < 32; g0 += G) {
		if (g0 + G < 32) {
#pragma unroll
			for (int j = 0; j < G; j++)
				nxt[j] = *reinterpret_cast<const uint4 *>(base + (g0 + G + j) * 256 + (LAZY ? offset(g0 + G + j) : off[g0 + G + j]));
		}
		asm volatile("" ::: "memory");
#pragma unroll
		for (int j = 0; j < G; j += 2) {
			acc.x = ct_xor3(acc.x, cur[j].x, cur[j + 1].x);
			acc.y = ct_xor3(acc.y, cur[j].y, cur[j + 1].y);
			acc.z = ct_xor3(acc.z, cur[j].z, cur[j + 1].z);
			acc.w = ct_xor3(acc.w, cur[j].w, cur[j + 1].w);
		}
		asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w)::"memory");
#pragma unroll
		for (int j = 0; j < G; j++)
			cur[j] = nxt[j];
	}
	return acc;
}

// A second nibble table that exists only in the kernel variants that need one (no LDS in the others)
template <bool ON>
struct ctable_opt {
	ctable_smem t;
	__device__ __forceinline__ ctable_smem &get() { return t; }
};
template <>
struct ctable_opt<false> {
};

} // namespace bn
