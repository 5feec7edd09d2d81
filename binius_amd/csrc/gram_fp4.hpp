// binius_amd/csrc/gram_fp4.hpp -- the GF(2) Gram products of gram.hpp on the FP4 matrix path: operand encoding, staging
// (nibble transpose), Gram k-steps and the parity read-out.  Used by kernels_roundeval_fp4.hip (round evaluation alone: every wave
// stages and runs k-steps, or -- whole tiles from 2^20 points -- stager waves and Gram waves) and by kernels_foldeval_fp4.hip (fused fold + evaluation: fold waves stage, Gram waves run the k-steps).
// The single-kind-of-wave fused kernel (kernels_foldeval_mfma.hip) runs its Gram part in int8 (gram.hpp): FP4 k-steps next to the
// constant multiplication in ONE wave did not fit the register file.  The design notes are at the top of
// kernels_roundeval_fp4.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "gram.hpp"

namespace bn {
namespace gram4 {

using namespace gram;

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int kT4W = 4096;         // words of the data part of a tile
constexpr int kTile4W = 4096 + 2048; // + the bit-3 words

struct stage4_role {
	uint32_t sel1, sel2, rot3, keep3;
	unsigned st_off;  // word offset of this lane's word inside a (set, limb) group: k-step, nibble index, point group
	unsigned st_off3; // the same inside a (set, limb pair) group of bit-3 words (rotated by half a block, see make_gram4_role)
};
// tid: index of the lane among the 256 staging lanes of the tile (the workgroup's threads in kernels_roundeval_fp4.hip; the
// fold waves' lanes in kernels_foldeval_fp4.hip)
__device__ __forceinline__ stage4_role make_stage4_role(unsigned tid)
{
	const unsigned j = tid & 7, jj = j < 4 ? j : 7 - j;
	stage4_role r;
	r.sel1 = j < 4 ? 0x05040100u : 0x03020706u;
	r.sel2 = (jj & 1) ? 0x03070105u : 0x06020400u;
	r.rot3 = (jj & 2) ? 4u : 28u;
	r.keep3 = (jj & 2) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
	const unsigned cidx = (j < 4 ? 0u : 4u) + ((jj & 1) << 1) + ((jj >> 1) & 1); // the nibble index lane j ends up holding
	const unsigned g = tid >> 3;
	// inside a 64-word block: 16-byte chunk (nibble index + 8 * k half), word = point group & 3.  The 32 lanes of a half wave
	// (eight nibble indices x four point groups) then write 32 consecutive banks; the bit-3 words sit four chunks further
	// round (mod 8), see make_gram4_role.
	const unsigned gq = g & 7;
	r.st_off = (g >> 3) * 64 + (cidx + 8 * (gq >> 2)) * 4 + (gq & 3);
	r.st_off3 = (g >> 3) * 64 + (((cidx + 4) & 7) + 8 * (gq >> 2)) * 4 + (gq & 3);
	return r;
}

__device__ __forceinline__ stage4_role make_stage4_role() { return make_stage4_role(threadIdx.x); }

// word x of this lane's point -> the word whose nibble i is nibble c(j) of x in the i-th lane (fixed order) of the
// group of eight; c(j) = [0, 2, 1, 3, 7, 5, 6, 4][j]
__device__ __forceinline__ uint32_t nib_tr(uint32_t x, const stage4_role &sr)
{
	const uint32_t p1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true); // row_half_mirror: lane 7 - j
	const uint32_t a = __builtin_amdgcn_perm(p1, x, sr.sel1);
	const uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
	const uint32_t b = __builtin_amdgcn_perm(p2, a, sr.sel2);
	const uint32_t p3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)b, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
	const uint32_t rot = __builtin_amdgcn_alignbit(p3, p3, sr.rot3);
	return __builtin_amdgcn_bitop3_b32(sr.keep3, b, rot, 0xCA); // (keep & b) | (~keep & rot)
}

// the four limbs of one element of set `set` -> four data words + two bit-3 words
__device__ __forceinline__ void stage4_elem(uint32_t *T, const stage4_role &sr, unsigned set, uint4 e)
{
	uint32_t *dst = T + set * 1024 + sr.st_off;
	const uint32_t y0 = nib_tr(e.x, sr), y1 = nib_tr(e.y, sr), y2 = nib_tr(e.z, sr), y3 = nib_tr(e.w, sr);
	dst[0 * 256] = y0;
	dst[1 * 256] = y1;
	dst[2 * 256] = y2;
	dst[3 * 256] = y3;
	uint32_t *dw = T + kT4W + set * 512 + sr.st_off3;
	// bit 3 of the even limb's nibbles at bit 2, of the odd limb's at bit 1 -- the only two bits the k-steps' masks keep of a bit-3
	// word (make_gram4_role: 0x44444444 / 0x22222222), so whatever else the select drags along is never read: three instructions
	dw[0] = __builtin_amdgcn_bitop3_b32(y0, y1 >> 1, 0x88888888u, 0xE4) >> 1; // ((y0 & m) | ((y1 >> 1) & ~m)) >> 1
	dw[256] = __builtin_amdgcn_bitop3_b32(y2, y3 >> 1, 0x88888888u, 0xE4) >> 1;
}

struct gram4_role {
	unsigned pr, h;
	unsigned u_off[4], v_off[2]; // word offsets of this lane's 16 bytes, k-step 0: u limbs 0..3, v limbs h and 2 + h
	uint32_t m_even, m_odd;      // masks for even / odd limbs (they differ only for the bit-3 rows)
	int e_row[2], e_col;         // exponent of this lane's column weight; of a register's row weight: see tail4
};
__device__ __forceinline__ gram4_role make_gram4_role(unsigned wave, unsigned lane)
{
	gram4_role g;
	g.pr = wave >> 1;
	g.h = wave & 1;
	const unsigned i = lane & 31, kh = lane >> 5, c = i >> 2, s = i & 3;
	// (a group of 16 lanes -- 12 data readers of four nibble indices, 4 bit-3 readers -- touches 8 distinct 16-byte chunks in 8
	// distinct bank groups: the bit-3 chunk of nibble index c sits where the data chunk of c + 4 would)
	const unsigned in_blk = s < 3 ? (c + 8 * kh) * 4 : (((c + 4) & 7) + 8 * kh) * 4;
	const unsigned us = 2 * g.pr, vs = 2 * g.pr + 1;
#pragma unroll
	for (unsigned w = 0; w < 4; w++)
		g.u_off[w] = s < 3 ? us * 1024 + w * 256 + in_blk : kT4W + us * 512 + (w >> 1) * 256 + in_blk;
#pragma unroll
	for (unsigned q = 0; q < 2; q++) {
		const unsigned w = g.h + 2 * q;
		g.v_off[q] = s < 3 ? vs * 1024 + w * 256 + in_blk : kT4W + vs * 512 + (w >> 1) * 256 + in_blk;
	}
	g.m_even = s < 3 ? 0x11111111u << s : 0x44444444u;
	g.m_odd = s < 3 ? 0x11111111u << s : 0x22222222u;
	g.e_col = s < 3 ? (int)s - 1 : (g.h ? 0 : 1);
	return g;
}

__device__ __forceinline__ void acc4_zero(v16f (&acc)[kAccTiles])
{
#pragma unroll
	for (int t = 0; t < kAccTiles; t++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[t][r] = 0.0f;
}

#define BN_GRAM4_MFMA(t, A, B)                                                                                                             \
	acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{(A).x, (A).y, (A).z, (A).w, 0, 0, 0, 0}, v8i{(B).x, (B).y, (B).z, (B).w, 0, 0, 0, 0}, \
	                                                         acc[t], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)

// One k-step (64 points): 6 ds_read_b128, 36 bitwise VALU, 6 MFMAs.  No second operand register set (three waves per SIMD
// at <= 168 registers hide the LDS latency of a k-step better than a prefetch at two waves does).
template <int KS>
__device__ __forceinline__ void gram4_step(const uint32_t *T, const gram4_role &g, v16f (&acc)[kAccTiles])
{
	v4i u[4], v[2];
#pragma unroll
	for (int w = 0; w < 4; w++)
		u[w] = *reinterpret_cast<const v4i *>(T + g.u_off[w] + KS * 64);
	v[0] = *reinterpret_cast<const v4i *>(T + g.v_off[0] + KS * 64);
	v[1] = *reinterpret_cast<const v4i *>(T + g.v_off[1] + KS * 64);
	const uint32_t me = g.m_even, mo = g.m_odd, mh = g.h ? mo : me;
	{
		const v4i B = and4(v[0], mh);
		BN_GRAM4_MFMA(0, and4(u[0], me), B);
		BN_GRAM4_MFMA(1, and4(u[1], mo), B);
	}
	{
		const v4i B = and4(v[1], mh);
		BN_GRAM4_MFMA(2, and4(u[2], me), B);
		BN_GRAM4_MFMA(3, and4(u[3], mo), B);
	}
	{
		const v4i B = xand4(v[0], v[1], mh);
		BN_GRAM4_MFMA(4, xand4(u[0], u[2], me), B);
		BN_GRAM4_MFMA(5, xand4(u[1], u[3], mo), B);
	}
	__builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void gram4_tile(const uint32_t *T, const gram4_role &g, v16f (&acc)[kAccTiles])
{
	gram4_step<0>(T, g, acc);
	gram4_step<1>(T, g, acc);
	gram4_step<2>(T, g, acc);
	gram4_step<3>(T, g, acc);
}

// ---- a second Karatsuba level: nine 32 x 32 blocks per product instead of twelve ------------------------------------------------
// u = (u0, u1, u2, u3) in 32-bit limbs.  Level 1 (GF(2^128) over GF(2^64)): terms s = 0: (u0, u1), 1: (u2, u3), 2: (u0 ^ u2,
// u1 ^ u3); level 2 (GF(2^64) over GF(2^32)) inside a term (a0, a1): t = 0: a0, 1: a1, 2: a0 ^ a1.  Block 3 s + t is the plain
// 32 x 32 Gram matrix of the (s, t) combination of u against the same combination of v.  The level-2 sums mix an even and an odd
// limb, so the bit-3 planes cannot share words the way stage4_elem packs them: here every limb has its own bit-3 word (bit 3 of
// a nibble at bit 2, whatever else the shift leaves is masked by the k-steps), a tile is 4096 + 4096 words.
// A product's nine blocks go to two Gram waves: q = 0 takes blocks 0 .. 4, q = 1 blocks 5 .. 8 (18 MFMAs per k-step and workgroup
// instead of 24; five or four accumulator tiles per wave instead of six, which is what leaves room for a second operand set).
// Used by k_roundeval_fp4_ws (kernels_roundeval_fp4.hip).  In the three-workgroup kernel the same blocks were measured no faster
// (profiles/r04/experiments/fp4_karatsuba2.txt): they pay through the software pipelining they make room for, not their MFMA count.
constexpr int kTile4kW = 4096 + 4096;

__device__ __forceinline__ void stage4k_elem(uint32_t *T, const stage4_role &sr, unsigned set, uint4 e)
{
	uint32_t *dst = T + set * 1024 + sr.st_off;
	const uint32_t y0 = nib_tr(e.x, sr), y1 = nib_tr(e.y, sr), y2 = nib_tr(e.z, sr), y3 = nib_tr(e.w, sr);
	dst[0 * 256] = y0;
	dst[1 * 256] = y1;
	dst[2 * 256] = y2;
	dst[3 * 256] = y3;
	uint32_t *dw = T + kT4W + set * 1024 + sr.st_off3;
	dw[0 * 256] = y0 >> 1;
	dw[1 * 256] = y1 >> 1;
	dw[2 * 256] = y2 >> 1;
	dw[3 * 256] = y3 >> 1;
}

struct gram4k_role {
	unsigned pr, q;
	unsigned u_off[4], v_off[4]; // word offsets of this lane's 16 bytes of limb w, k-step 0
	uint32_t m;                  // this lane's bit of every nibble (data rows) / bit 2 of the bit-3 words
	int e;                       // exponent of the FP4 value that bit decodes to
};
__device__ __forceinline__ gram4k_role make_gram4k_role(unsigned pr, unsigned q, unsigned lane)
{
	gram4k_role g;
	g.pr = pr;
	g.q = q;
	const unsigned i = lane & 31, kh = lane >> 5, c = i >> 2, s = i & 3;
	const unsigned in_blk = s < 3 ? (c + 8 * kh) * 4 : (((c + 4) & 7) + 8 * kh) * 4; // (as make_gram4_role)
	const unsigned base = s < 3 ? 0u : (unsigned)kT4W;
#pragma unroll
	for (unsigned w = 0; w < 4; w++) {
		g.u_off[w] = base + (2 * pr) * 1024 + w * 256 + in_blk;
		g.v_off[w] = base + (2 * pr + 1) * 1024 + w * 256 + in_blk;
	}
	g.m = s < 3 ? 0x11111111u << s : 0x44444444u;
	g.e = s < 3 ? (int)s - 1 : 1;
	return g;
}
__device__ __forceinline__ v4i x4and4(v4i a, v4i b, v4i c, v4i d, uint32_t m) // (a ^ b ^ c ^ d) & m
{
	return xand4(xor3_4(a, b, c), d, m);
}
#define BN_GRAM4K_MFMA(t, A, B)                                                                                                            \
	acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{(A).x, (A).y, (A).z, (A).w, 0, 0, 0, 0}, v8i{(B).x, (B).y, (B).z, (B).w, 0, 0, 0, 0}, \
	                                                         acc[t], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)
// Q = 0: blocks 0 .. 4 (five accumulator tiles), Q = 1: blocks 5 .. 8 (four).
// N k-steps in a row (step ks reads tile ks / 4 of a run of tiles kTile4kW words apart), software-pipelined for a Gram wave that has
// the SIMD's matrix pipe to itself (kernels_roundeval_fp4.hip, wave-specialised form): two operand register sets of four limbs each,
// every set refilled for its next use as soon as the masks that read it are formed, so that a read has a whole group of MFMAs to land in.
template <int Q, int N>
__device__ __forceinline__ void gram4k_steps(const uint32_t *T, const gram4k_role &g, v16f (&acc)[Q == 0 ? 5 : 4])
{
	const uint32_t m = g.m;
	auto at = [&](unsigned off, int ks) { return *reinterpret_cast<const v4i *>(T + off + (ks >> 2) * kTile4kW + (ks & 3) * 64); };
	if constexpr (Q == 0) {
		// set A = (u0, u1, v0, v1) -> blocks 0, 1, 2; set B = (u2, u3, v2, v3) -> blocks 3, 4
		v4i a0 = at(g.u_off[0], 0), a1 = at(g.u_off[1], 0), a2 = at(g.v_off[0], 0), a3 = at(g.v_off[1], 0);
		v4i b0 = at(g.u_off[2], 0), b1 = at(g.u_off[3], 0), b2 = at(g.v_off[2], 0), b3 = at(g.v_off[3], 0);
#pragma unroll
		for (int ks = 0; ks < N; ks++) {
			{
				const v4i A0 = and4(a0, m), B0 = and4(a2, m), A1 = and4(a1, m), B1 = and4(a3, m);
				const v4i A2 = xand4(a0, a1, m), B2 = xand4(a2, a3, m);
				if (ks + 1 < N) {
					a0 = at(g.u_off[0], ks + 1);
					a1 = at(g.u_off[1], ks + 1);
					a2 = at(g.v_off[0], ks + 1);
					a3 = at(g.v_off[1], ks + 1);
				}
				BN_GRAM4K_MFMA(0, A0, B0);
				BN_GRAM4K_MFMA(1, A1, B1);
				BN_GRAM4K_MFMA(2, A2, B2);
			}
			__builtin_amdgcn_sched_barrier(0);
			{
				const v4i A3 = and4(b0, m), B3 = and4(b2, m), A4 = and4(b1, m), B4 = and4(b3, m);
				if (ks + 1 < N) {
					b0 = at(g.u_off[2], ks + 1);
					b1 = at(g.u_off[3], ks + 1);
					b2 = at(g.v_off[2], ks + 1);
					b3 = at(g.v_off[3], ks + 1);
				}
				BN_GRAM4K_MFMA(3, A3, B3);
				BN_GRAM4K_MFMA(4, A4, B4);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	} else {
		// set A = the four u limbs, set B = the four v limbs; blocks 5 .. 8 = (2^3, 0^2, 1^3, 0^1^2^3) of both
		v4i a0 = at(g.u_off[0], 0), a1 = at(g.u_off[1], 0), a2 = at(g.u_off[2], 0), a3 = at(g.u_off[3], 0);
		v4i b0 = at(g.v_off[0], 0), b1 = at(g.v_off[1], 0), b2 = at(g.v_off[2], 0), b3 = at(g.v_off[3], 0);
#pragma unroll
		for (int ks = 0; ks < N; ks++) {
			const v4i A5 = xand4(a2, a3, m), A6 = xand4(a0, a2, m), A7 = xand4(a1, a3, m), A8 = x4and4(a0, a1, a2, a3, m);
			if (ks + 1 < N) {
				a0 = at(g.u_off[0], ks + 1);
				a1 = at(g.u_off[1], ks + 1);
				a2 = at(g.u_off[2], ks + 1);
				a3 = at(g.u_off[3], ks + 1);
			}
			__builtin_amdgcn_sched_barrier(0);
			{
				const v4i B5 = xand4(b2, b3, m), B6 = xand4(b0, b2, m);
				BN_GRAM4K_MFMA(0, A5, B5);
				BN_GRAM4K_MFMA(1, A6, B6);
				const v4i B7 = xand4(b1, b3, m), B8 = x4and4(b0, b1, b2, b3, m);
				if (ks + 1 < N) {
					b0 = at(g.v_off[0], ks + 1);
					b1 = at(g.v_off[1], ks + 1);
					b2 = at(g.v_off[2], ks + 1);
					b3 = at(g.v_off[3], ks + 1);
				}
				BN_GRAM4K_MFMA(2, A7, B7);
				BN_GRAM4K_MFMA(3, A8, B8);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// parity words of a wave's blocks: P[product][block][lane] (register r of a tile is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5); its
// weight's exponent depends on r & 3 only)
typedef uint32_t gram4k_parity[2][9][64];
template <int Q>
__device__ __forceinline__ void parity4k(const v16f (&acc)[Q == 0 ? 5 : 4], const gram4k_role &g, unsigned lane, gram4k_parity &P)
{
#pragma unroll
	for (int j = 0; j < (Q == 0 ? 5 : 4); j++) {
		uint32_t v = 0;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int sr = r & 3;
			const int e_row = sr < 3 ? sr - 1 : 1;
			const int cnt = (int)__builtin_amdgcn_ldexpf(acc[j][r], -(e_row + g.e));
			v |= ((uint32_t)cnt & 1u) << r;
		}
		P[g.pr][5 * Q + j][lane] = v;
	}
}
// the eighteen blocks -> the six GF(2^64) sums tail_publish starts from.  All threads of the workgroup; P complete behind the barrier in here.
__device__ __forceinline__ void sums4k(gram4k_parity &P, unsigned wave, unsigned lane, uint64_t (&z3)[2][3])
{
	__shared__ uint32_t z9[2][9];
	__syncthreads();
	const unsigned n = lane & 31;
	// block (pr, b): column n as a GF(2^32) element (bit p = G[p][n]); z = sum_n col * 2^n in GF(2^32); one block per half wave
	for (unsigned task = 2 * wave + (lane >> 5); task < 18; task += 2 * (blockDim.x >> 6)) {
		const unsigned pr = task / 9, b = task - 9 * pr;
		auto spread = [](uint32_t x) { return (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12); };
		const uint32_t *gp = P[pr][b];
		uint64_t z = spread(gp[n]) | (spread(gp[n + 32]) << 4);
		if (n & 1) z = mulx64<0>(z);
		if (n & 2) z = mulx64<1>(z);
		if (n & 4) z = mulx64<2>(z);
		if (n & 8) z = mulx64<3>(z);
		if (n & 16) z = mulx64<4>(z);
		uint32_t zz = (uint32_t)z;
#pragma unroll
		for (int mm = 16; mm >= 1; mm >>= 1)
			zz ^= (uint32_t)__shfl_xor((int)zz, mm, 64);
		if (n == 0) z9[pr][b] = zz;
	}
	__syncthreads();
	if (threadIdx.x < 6) {
		// term s of product pr: (t0, t1, t2) = (lo lo, hi hi, middle) in GF(2^32) -> lo = t0 + t1 ; hi = t2 + t0 + t1 + t1 X_4
		const unsigned pr = threadIdx.x / 3, s = threadIdx.x - 3 * pr;
		const uint32_t t0 = z9[pr][3 * s], t1 = z9[pr][3 * s + 1], t2 = z9[pr][3 * s + 2];
		const uint32_t lo = t0 ^ t1;
		const uint32_t hi = t2 ^ lo ^ (uint32_t)mulx64<4>((uint64_t)t1);
		z3[pr][s] = (uint64_t)lo | ((uint64_t)hi << 32);
	}
}

// parity bits out of the f32 accumulators, then the common tail.  Register r of a tile is row (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5): its bit position inside the nibble is r & 3; tile t = 2 s + i has rows from an even (i = 0) or odd limb.
__device__ __forceinline__ void parity4(const v16f (&acc)[kAccTiles], const gram4_role &g, unsigned wave, unsigned lane, gram_parity &Gc)
{
#pragma unroll
	for (int t = 0; t < kAccTiles; t++) {
		uint32_t v = 0;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int sr = r & 3;
			const int e_row = sr < 3 ? sr - 1 : ((t & 1) ? 0 : 1);
			const int cnt = (int)__builtin_amdgcn_ldexpf(acc[t][r], -(e_row + g.e_col));
			v |= ((uint32_t)cnt & 1u) << r;
		}
		Gc[wave][t][lane] = v;
	}
}
__device__ __forceinline__ void tail4(const v16f (&acc)[kAccTiles], const gram4_role &g, unsigned wave, unsigned lane, f128 *out,
                                      const fin_fuse &fz, uint64_t seq, const fin_cache *fc = nullptr)
{
	__shared__ gram_parity Gc;
	parity4(acc, g, wave, lane, Gc);
	tail_finish(Gc, wave, lane, out, fz, seq, fc);
}


} // namespace gram4
} // namespace bn
