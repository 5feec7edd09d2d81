// binius_amd/csrc/bitslice.hpp -- bit-sliced binary-tower arithmetic for gfx950 wavefronts.
//
// A "plane set" of level K is 2^K 32-bit registers per lane; register p holds bit p of 32
// independent field elements (one element per bit position).  AND/XOR on registers then act on 32
// elements at once, which is the only way to get variable x variable GF(2^128) products onto a GPU
// with no carry-less multiply: the tower recursion of
// crates/field/src/arch/portable/pairwise_recursive_arithmetic.rs:18-28 becomes ~13k AND/XOR per
// 32 products (v_bitop3_b32 fuses AND+XOR pairs), i.e. ~400 VALU lane-ops per product instead of
// the ~6000 of any word-level formulation.
//
// Bit positions inside a register are opaque to the arithmetic, so a register may equally hold
// two 16-element groups (low / high half) that are multiplied "in parallel" -- the round-eval
// kernel packs the evaluation-at-1 operands in the low half and the evaluation-at-infinity
// operands in the high half.
#pragma once
#include <stdint.h>

#include "gf128.hpp"

namespace bn {

// ---- 32x32 bit-matrix transpose, in registers --------------------------------------------------
// in : r[e] = 32-bit word of element e            (row e, column = bit index)
// out: r[p] = plane p, bit e = bit p of element e
// (a & m) | (b & ~m) in one v_bitop3_b32 (truth table: src0=0xF0, src1=0xCC, src2=0xAA)
__device__ __forceinline__ uint32_t bitsel(uint32_t a, uint32_t b, uint32_t m)
{
	return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4);
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
	return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// (a & b) ^ c in one v_bitop3_b32
__device__ __forceinline__ uint32_t andxor(uint32_t a, uint32_t b, uint32_t c)
{
	return __builtin_amdgcn_bitop3_b32(a, b, c, 0x6A);
}

__device__ __forceinline__ void transpose32(uint32_t (&r)[32])
{
	// j = 16 and j = 8 move whole bytes: one v_perm_b32 per output (selector byte k picks byte k of
	// the result from {src0 = bytes 7..4, src1 = bytes 3..0})
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const uint32_t a = r[k], b = r[k + 16];
		r[k] = __builtin_amdgcn_perm(b, a, 0x05040100u);      // (a & 0xFFFF) | (b << 16)
		r[k + 16] = __builtin_amdgcn_perm(b, a, 0x07060302u); // (a >> 16) | (b & 0xFFFF0000)
	}
#pragma unroll
	for (int k = 0; k < 32; k++) {
		if (k & 8) continue;
		const uint32_t a = r[k], b = r[k + 8];
		r[k] = __builtin_amdgcn_perm(b, a, 0x06020400u);     // bytes a0 b0 a2 b2
		r[k + 8] = __builtin_amdgcn_perm(b, a, 0x07030501u); // bytes a1 b1 a3 b3
	}
	// j = 4, 2, 1: shift + one 3-input bit select per output
#pragma unroll
	for (int k = 0; k < 32; k++) {
		if (k & 4) continue;
		const uint32_t a = r[k], b = r[k + 4];
		r[k] = bitsel(a, b << 4, 0x0F0F0F0Fu);
		r[k + 4] = bitsel(a >> 4, b, 0x0F0F0F0Fu);
	}
#pragma unroll
	for (int k = 0; k < 32; k++) {
		if (k & 2) continue;
		const uint32_t a = r[k], b = r[k + 2];
		r[k] = bitsel(a, b << 2, 0x33333333u);
		r[k + 2] = bitsel(a >> 2, b, 0x33333333u);
	}
#pragma unroll
	for (int k = 0; k < 32; k++) {
		if (k & 1) continue;
		const uint32_t a = r[k], b = r[k + 1];
		r[k] = bitsel(a, b << 1, 0x55555555u);
		r[k + 1] = bitsel(a >> 1, b, 0x55555555u);
	}
}

// ---- bit-sliced tower arithmetic ---------------------------------------------------------------
// All functions take pointers into register arrays; after full unrolling every index is a
// compile-time constant, so nothing touches scratch.

// t = a * X_{K-1} for a level-K plane set: (a0, a1) -> (a1, a0 + a1 * X_{K-2})   (mul_alpha,
// pairwise_recursive_arithmetic.rs:54-60).  Pure XOR / renaming.  out must not alias a.
template <int K>
__device__ __forceinline__ void bs_mul_alpha(const uint32_t *a, uint32_t *out)
{
	if constexpr (K == 0) {
		out[0] = a[0];
	} else {
		constexpr int H = 1 << (K - 1);
		uint32_t t[H];
		bs_mul_alpha<K - 1>(a + H, t);
#pragma unroll
		for (int i = 0; i < H; i++) {
			out[i] = a[H + i];
			out[H + i] = a[i] ^ t[i];
		}
	}
}

// out = a * b, level K (2^K planes each).  out must not alias a or b.
// LUT = true writes the two lowest levels out on the 3-input LUT (880 instructions at K = 5); LUT = false keeps the
// Karatsuba recursion down to GF(4) (1015 instructions, but shorter live ranges: the and-xor chains of the LUT form
// keep all eight inputs of a GF(16) product alive to the end, which is what tips the fused fold + evaluate kernels of
// kernels_foldeval9.hip -- already at the 256-register limit -- into scratch: 0 -> 242 spilled registers).
template <int K, bool LUT = true>
__device__ __forceinline__ void bs_mul(const uint32_t *a, const uint32_t *b, uint32_t *out)
{
	if constexpr (K == 0) {
		out[0] = a[0] & b[0];
	} else if constexpr (K == 1 && !LUT) {
		// GF(4): lo = a0b0 ^ a1b1 ; hi = (a0^a1)(b0^b1) ^ a0b0      (alpha_0 = 1)
		uint32_t z0 = a[0] & b[0];
		out[0] = (a[1] & b[1]) ^ z0;
		out[1] = ((a[0] ^ a[1]) & (b[0] ^ b[1])) ^ z0;
	} else if constexpr (K == 1) {
		// GF(4), alpha_0 = 1: lo = a0b0 ^ a1b1 ; hi = a0b1 ^ a1b0 ^ a1b1.  Schoolbook on the 3-input LUT: every
		// monomial after the first is one (x & y) ^ acc, 4 ops against 5 for the Karatsuba form.
		const uint32_t t = a[1] & b[1];
		out[0] = andxor(a[0], b[0], t);
		out[1] = andxor(a[1], b[0], andxor(a[0], b[1], t));
	} else if constexpr (K == 2 && LUT) {
		// GF(16) = GF(4)[X]/(X^2 + X*X_0 + 1), written out over the bits: 19 ops against 24 for a Karatsuba level over
		// three GF(4) products (the pre-additions disappear, the recombination rides in the and-xor chains).
		//   z  = a_hi * b_hi                              (GF(4), 4 ops)
		//   lo = a_lo * b_lo ^ z                          (chains seeded with z)
		//   hi = a_lo * b_hi ^ a_hi * b_lo ^ z * X_0,     z * X_0 = (z1, z0 ^ z1)
		const uint32_t t = a[3] & b[3];
		const uint32_t z0 = andxor(a[2], b[2], t);
		const uint32_t z1 = andxor(a[3], b[2], andxor(a[2], b[3], t));
		out[0] = andxor(a[1], b[1], andxor(a[0], b[0], z0));
		out[1] = andxor(a[1], b[1], andxor(a[1], b[0], andxor(a[0], b[1], z1)));
		// a1b3 ^ a3b1 is common to both high bits
		const uint32_t s0 = andxor(a[3], b[1], a[1] & b[3]) ^ z1;
		out[2] = andxor(a[2], b[0], andxor(a[0], b[2], s0));
		out[3] = andxor(a[3], b[0], andxor(a[2], b[1], andxor(a[1], b[2], andxor(a[0], b[3], s0 ^ z0))));
	} else {
		constexpr int H = 1 << (K - 1);
		uint32_t z0[H], z2[H], z1[H], sa[H], sb[H], za[H];
		bs_mul<K - 1, LUT>(a, b, z0);
		bs_mul<K - 1, LUT>(a + H, b + H, z2);
#pragma unroll
		for (int i = 0; i < H; i++) {
			sa[i] = a[i] ^ a[H + i];
			sb[i] = b[i] ^ b[H + i];
		}
		bs_mul<K - 1, LUT>(sa, sb, z1);
		bs_mul_alpha<K - 1>(z2, za);
#pragma unroll
		for (int i = 0; i < H; i++) {
			uint32_t lo = z0[i] ^ z2[i];
			out[i] = lo;
			out[H + i] = xor3(z1[i], lo, za[i]); // one v_bitop3_b32
		}
	}
}

// out = a * b at level K with the three half-size products of every level >= SEQ issued strictly one after the other
// (scheduling barriers in between) and the operands of the middle product formed only when it is their turn.  The
// plain recursion above leaves ~1000 independent AND/XOR to the scheduler, which interleaves the sub-products and keeps
// ~200 temporaries alive; sequenced, a level-5 product needs its operands, its result and ~60 temporaries.
// out must not alias a or b.
template <int K, int SEQ = 4>
__device__ __forceinline__ void bs_mul_seq(const uint32_t *a, const uint32_t *b, uint32_t *out)
{
	if constexpr (K < SEQ) {
		bs_mul<K>(a, b, out);
	} else {
		constexpr int H = 1 << (K - 1);
		uint32_t z[H];
		bs_mul_seq<K - 1, SEQ>(a, b, z); // z0
#pragma unroll
		for (int i = 0; i < H; i++) {
			out[i] = z[i];
			out[H + i] = z[i];
		}
		__builtin_amdgcn_sched_barrier(0);
		{
			uint32_t z2[H], za[H];
			bs_mul_seq<K - 1, SEQ>(a + H, b + H, z2);
			bs_mul_alpha<K - 1>(z2, za);
#pragma unroll
			for (int i = 0; i < H; i++) {
				out[i] ^= z2[i];
				out[H + i] = xor3(out[H + i], z2[i], za[i]);
			}
		}
		__builtin_amdgcn_sched_barrier(0);
		{
			uint32_t sa[H], sb[H];
#pragma unroll
			for (int i = 0; i < H; i++) {
				sa[i] = a[i] ^ a[H + i];
				sb[i] = b[i] ^ b[H + i];
			}
			bs_mul_seq<K - 1, SEQ>(sa, sb, z); // z1
#pragma unroll
			for (int i = 0; i < H; i++)
				out[H + i] ^= z[i];
		}
		__builtin_amdgcn_sched_barrier(0);
	}
}

// acc ^= a * b at level K, where acc has 2^K planes.
template <int K>
__device__ __forceinline__ void bs_mac(const uint32_t *a, const uint32_t *b, uint32_t *acc)
{
	constexpr int N = 1 << K;
	uint32_t p[N];
	bs_mul<K>(a, b, p);
#pragma unroll
	for (int i = 0; i < N; i++)
		acc[i] ^= p[i];
}

} // namespace bn
