// binius_amd/csrc/gram.hpp -- sums of GF(2^128) products on the matrix cores.
//
// The round evaluation of the bivariate product needs  S = sum_j u_j * v_j  over 2^k pairs
// (crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-408).  Multiplication is GF(2)-bilinear,
//     u * v = sum_{p,q} u[p] v[q] (e_p e_q)            e_i = basis element 2^i of the tower,
// so the sum over j is a function of the 128 x 128 GF(2) Gram matrix of the bit vectors only:
//     S = sum_{p,q} G[p][q] (e_p e_q),     G[p][q] = sum_j u_j[p] v_j[q]  (mod 2)  =  (U^T V)[p][q].
// That is a genuine contraction over j -- an integer matrix product whose entries are only needed
// modulo 2 -- and it is what `v_mfma_i32_32x32x32_i8` computes.  The bit-sliced formulation of round 1
// (bitslice.hpp) spends ~290 VALU lane-operations per product; here the products themselves cost no
// VALU at all, and what is left is operand preparation: ~2.5 lane-operations per product and byte.
//
//  * Operand bytes are the DATA BITS LEFT IN PLACE.  A staged word T[w][c] holds byte c of limb w
//    (32-bit word of the element) of four consecutive points; the MFMA row (or column) index is the bit
//    m = 8c + s of the limb, and lane m prepares its operand with ONE `v_and_b32` per register:
//    T & (0x01010101 << s).  A byte is then 0 or 2^s, a product 0 or 2^(s+t), and the i32 accumulator
//    of entry (m, n) counts in units of 2^(s+t): its bit (s + t) is the parity wanted, whatever wrapped
//    around above it (int8 0x80 = -128 only flips signs, which parity does not see).
//  * One Karatsuba level over the two 64-bit halves (pairwise_recursive_arithmetic.rs:18-28): three
//    64 x 64 Gram matrices (lo x lo, hi x hi, (lo+hi) x (lo+hi)) = twelve 32 x 32 tiles instead of sixteen;
//    every post-processing step is linear, commutes with the sum over j and happens once per
//    workgroup.  The operand of a "mid" tile is one `v_bitop3_b32` per register ((x ^ y) & mask).
//    A second level (nine tiles) was measured and is slower: every tile then needs two freshly masked
//    operands, and a SIMD issues ~12 bitwise VALU in the shadow of one MFMA before VALU issue, not the
//    matrix pipe, is the bound (tools/mfma_issue.hip).
//  * A wave owns one product (evaluation at 1, or at infinity) and one 32-bit column half h of each
//    64-bit half of v: six accumulator tiles = 96 registers, 6 MFMAs per 36 VALU and 6 ds_read_b128.
//
// LDS tile layout ("T"), 16 KiB per 256 points: [set 0..3][limb 0..3][k-step 0..7] blocks of 32 words; inside a
// block word 8c + q holds byte c of the limb for the four consecutive points of quad q (q = 0..7) of the
// k-step's 32 points.  A reader lane (m, kb) takes the 16 bytes at word 8 (m >> 3) + 4 kb: one
// ds_read_b128 per operand and limb, 8 lanes broadcasting each 16-byte chunk.
// Staging: lane = point.  A lane holds the four limbs of its element; the 4 x 4 byte transpose across the
// four lanes of a quad is two DPP exchanges (lane ^ 1, lane ^ 2) + two v_perm_b32 with per-lane selectors,
// after which lane c of the quad holds byte c of the limb for the quad's four points -- exactly one T
// word.  No LDS round trip and no barrier between the producer of an element and its T words.
// Sets: 0 = u at 1 (a_hi), 1 = v at 1 (b_hi), 2 = u at infinity (a_lo + a_hi), 3 = v at infinity.
#pragma once
#include <hip/hip_runtime.h>

#include "finalize.hpp"
#include "internal.hpp"

namespace bn {
namespace gram {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kTP = 256;             // points per tile = per workgroup iteration
constexpr int kBlkW = 32;            // words per (set, limb, k-step) block
constexpr int kLimbW = 8 * kBlkW;    // one limb of one set: 8 k-steps
constexpr int kSetW = 4 * kLimbW;    // one operand set of a tile
constexpr int kTileW = 4 * kSetW;    // 4096 words = 16 KiB

// Staging role of a lane: its wave stages the 64 points (k-steps 2 wave, 2 wave + 1) it loaded or folded.
struct stage_role {
	uint32_t sel_a, sel_b; // v_perm selectors of the two exchange stages
	unsigned st_off;       // word offset of this lane's T word inside a (set, limb) group of blocks
};
__device__ __forceinline__ stage_role make_stage_role(unsigned wave, unsigned lane)
{
	stage_role r;
	r.sel_a = (lane & 1) ? 0x03070105u : 0x06020400u;
	r.sel_b = (lane & 2) ? 0x03020706u : 0x05040100u;
	const unsigned q4 = lane >> 2, c = lane & 3;
	r.st_off = (2 * wave + (q4 >> 3)) * kBlkW + c * 8 + (q4 & 7);
	return r;
}

// word x of this lane's point -> the word whose byte k is byte (lane & 3) of x in lane k of the quad
__device__ __forceinline__ uint32_t quad_btr(uint32_t x, const stage_role &sr)
{
	const uint32_t p1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); // quad_perm [1,0,3,2]
	const uint32_t a = __builtin_amdgcn_perm(p1, x, sr.sel_a);
	const uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)a, 0x4E, 0xF, 0xF, true); // quad_perm [2,3,0,1]
	return __builtin_amdgcn_perm(p2, a, sr.sel_b);
}

// One operand o (0: a -> sets 0 and 2, 1: b -> sets 1 and 3) of this lane's point: hi = the element of the
// evaluation at 1, lo = its partner.  MIX: the second set holds hi ^ lo (evaluation at infinity);
// otherwise lo itself (two independent products).
template <bool MIX>
__device__ __forceinline__ void stage_T(uint32_t *T, const stage_role &sr, unsigned o, uint4 hi, uint4 lo)
{
	uint32_t *dst_hi = T + o * kSetW + sr.st_off;
	uint32_t *dst_mx = dst_hi + 2 * kSetW;
	const uint4 mx = MIX ? uint4{hi.x ^ lo.x, hi.y ^ lo.y, hi.z ^ lo.z, hi.w ^ lo.w} : lo;
	dst_hi[0 * kLimbW] = quad_btr(hi.x, sr);
	dst_hi[1 * kLimbW] = quad_btr(hi.y, sr);
	dst_hi[2 * kLimbW] = quad_btr(hi.z, sr);
	dst_hi[3 * kLimbW] = quad_btr(hi.w, sr);
	dst_mx[0 * kLimbW] = quad_btr(mx.x, sr);
	dst_mx[1 * kLimbW] = quad_btr(mx.y, sr);
	dst_mx[2 * kLimbW] = quad_btr(mx.z, sr);
	dst_mx[3 * kLimbW] = quad_btr(mx.w, sr);
}

// One limb of stage_T: piece (o, w) -- two T words.  Lets a caller spread the staging of a tile over the
// Gram k-steps of the previous one.
template <bool MIX, int W>
__device__ __forceinline__ void stage_T_limb(uint32_t *T, const stage_role &sr, unsigned o, const uint4 &hi, const uint4 &lo)
{
	const uint32_t h = W == 0 ? hi.x : (W == 1 ? hi.y : (W == 2 ? hi.z : hi.w));
	const uint32_t l = W == 0 ? lo.x : (W == 1 ? lo.y : (W == 2 ? lo.z : lo.w));
	uint32_t *dst_hi = T + o * kSetW + sr.st_off + W * kLimbW;
	dst_hi[0] = quad_btr(h, sr);
	dst_hi[2 * kSetW] = quad_btr(MIX ? (h ^ l) : l, sr);
}

__device__ __forceinline__ v4i and4(v4i x, uint32_t m)
{
	return v4i{(int)((uint32_t)x.x & m), (int)((uint32_t)x.y & m), (int)((uint32_t)x.z & m), (int)((uint32_t)x.w & m)};
}
__device__ __forceinline__ v4i xand4(v4i x, v4i y, uint32_t m) // (x ^ y) & m
{
	return v4i{(int)__builtin_amdgcn_bitop3_b32((uint32_t)x.x, (uint32_t)y.x, m, 0x28), (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.y, (uint32_t)y.y, m, 0x28),
	           (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.z, (uint32_t)y.z, m, 0x28), (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.w, (uint32_t)y.w, m, 0x28)};
}
__device__ __forceinline__ v4i xor3_4(v4i x, v4i y, v4i z)
{
	return v4i{(int)__builtin_amdgcn_bitop3_b32((uint32_t)x.x, (uint32_t)y.x, (uint32_t)z.x, 0x96), (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.y, (uint32_t)y.y, (uint32_t)z.y, 0x96),
	           (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.z, (uint32_t)y.z, (uint32_t)z.z, 0x96), (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.w, (uint32_t)y.w, (uint32_t)z.w, 0x96)};
}

// Compute role of a wave: product pr = wave >> 1, column half h = wave & 1; lane (m, kb).
struct gram_role {
	unsigned pr, h;
	unsigned u_off, v_off; // word offsets of this lane's 16 bytes in limb 0 of k-step 0 (u set / v set)
	uint32_t msk;
};
__device__ __forceinline__ gram_role make_gram_role(unsigned wave, unsigned lane)
{
	gram_role g;
	g.pr = wave >> 1;
	g.h = wave & 1;
	const unsigned m = lane & 31, kb = lane >> 5;
	g.u_off = (2 * g.pr) * kSetW + (m >> 3) * 8 + kb * 4;
	g.v_off = g.u_off + kSetW + g.h * kLimbW;
	g.msk = 0x01010101u << (m & 7);
	return g;
}

constexpr int kAccTiles = 6;

__device__ __forceinline__ void acc_zero(v16i (&acc)[kAccTiles])
{
#pragma unroll
	for (int t = 0; t < kAccTiles; t++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[t][r] = 0;
}

#define BN_GRAM_MFMA(t, A, B) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[t], 0, 0, 0)

// The eight k-steps (32 points each) of a tile: per k-step 6 ds_read_b128 (the four limbs of u, limbs h and
// 2 + h of v), 36 bitwise VALU and 6 MFMAs.  The limbs of k-step ks + 1 are requested before the MFMAs of
// k-step ks are issued (second register set).  The stepping form lets a caller put other work between
// k-steps (the fused kernel folds the next tile there).
struct gram_pipe {
	v4i u[4], v[2];
};
__device__ __forceinline__ void gram_begin(const uint32_t *T, const gram_role &g, gram_pipe &p)
{
	const uint32_t *U = T + g.u_off, *V = T + g.v_off;
#pragma unroll
	for (int w = 0; w < 4; w++)
		p.u[w] = *reinterpret_cast<const v4i *>(U + w * kLimbW);
	p.v[0] = *reinterpret_cast<const v4i *>(V);
	p.v[1] = *reinterpret_cast<const v4i *>(V + 2 * kLimbW);
}
template <int KS>
__device__ __forceinline__ void gram_step(const uint32_t *T, const gram_role &g, gram_pipe &p, v16i (&acc)[kAccTiles])
{
	const uint32_t *U = T + g.u_off, *V = T + g.v_off;
	const uint32_t msk = g.msk;
	v4i un[4], vn[2];
	if (KS < 7) {
#pragma unroll
		for (int w = 0; w < 4; w++)
			un[w] = *reinterpret_cast<const v4i *>(U + w * kLimbW + (KS + 1) * kBlkW);
		vn[0] = *reinterpret_cast<const v4i *>(V + (KS + 1) * kBlkW);
		vn[1] = *reinterpret_cast<const v4i *>(V + 2 * kLimbW + (KS + 1) * kBlkW);
		// keep the requests HERE, ahead of this k-step's MFMAs: left alone, the scheduler sinks them to
		// their first use (or hoists the next k-step's operand preparation up to them) and the LDS latency
		// of every k-step is exposed
		asm volatile("" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
	}
#ifndef GRAM_VARIANT
#define GRAM_VARIANT 0
#endif
#if GRAM_VARIANT == 0
	{
		const v4i B = and4(p.v[0], msk);
		BN_GRAM_MFMA(0, and4(p.u[0], msk), B);
		BN_GRAM_MFMA(1, and4(p.u[1], msk), B);
	}
	{
		const v4i B = and4(p.v[1], msk);
		BN_GRAM_MFMA(2, and4(p.u[2], msk), B);
		BN_GRAM_MFMA(3, and4(p.u[3], msk), B);
	}
	{
		const v4i B = xand4(p.v[0], p.v[1], msk);
		BN_GRAM_MFMA(4, xand4(p.u[0], p.u[2], msk), B);
		BN_GRAM_MFMA(5, xand4(p.u[1], p.u[3], msk), B);
	}
#else
	{
		// all nine operands first (distinct registers), then the six MFMAs back to back
		const v4i B0 = and4(p.v[0], msk), B1 = and4(p.v[1], msk), B2 = xand4(p.v[0], p.v[1], msk);
		const v4i A0 = and4(p.u[0], msk), A1 = and4(p.u[1], msk), A2 = and4(p.u[2], msk), A3 = and4(p.u[3], msk);
		const v4i A4 = xand4(p.u[0], p.u[2], msk), A5 = xand4(p.u[1], p.u[3], msk);
		__builtin_amdgcn_sched_barrier(0);
		BN_GRAM_MFMA(0, A0, B0);
		BN_GRAM_MFMA(1, A1, B0);
		BN_GRAM_MFMA(2, A2, B1);
		BN_GRAM_MFMA(3, A3, B1);
		BN_GRAM_MFMA(4, A4, B2);
		BN_GRAM_MFMA(5, A5, B2);
		__builtin_amdgcn_sched_barrier(0);
	}
#endif
	if (KS < 7) {
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int w = 0; w < 4; w++)
			p.u[w] = un[w];
		p.v[0] = vn[0];
		p.v[1] = vn[1];
	}
}
__device__ __forceinline__ void gram_tile(const uint32_t *T, const gram_role &g, v16i (&acc)[kAccTiles])
{
	gram_pipe p;
	gram_begin(T, g, p);
	gram_step<0>(T, g, p, acc);
	gram_step<1>(T, g, p, acc);
	gram_step<2>(T, g, p, acc);
	gram_step<3>(T, g, p, acc);
	gram_step<4>(T, g, p, acc);
	gram_step<5>(T, g, p, acc);
	gram_step<6>(T, g, p, acc);
	gram_step<7>(T, g, p, acc);
}

__device__ __forceinline__ uint64_t mul_basis64(uint64_t z, unsigned i) // z * 2^i in GF(2^64), i < 64
{
	if (i & 1) z = mulx64<0>(z);
	if (i & 2) z = mulx64<1>(z);
	if (i & 4) z = mulx64<2>(z);
	if (i & 8) z = mulx64<3>(z);
	if (i & 16) z = mulx64<4>(z);
	if (i & 32) z = mulx64<5>(z);
	return z;
}

// (z0, z2, z1') of the Karatsuba level -> the product: lo = z0 + z2 ; hi = z1' + z0 + z2 + z2 * X_5
__device__ __forceinline__ f128 kara64(uint64_t Z0, uint64_t Z2, uint64_t Z1p)
{
	const uint64_t lo = Z0 ^ Z2;
	return f128{lo, Z1p ^ lo ^ mulx64<5>(Z2)};
}

// Workgroup tail (all 256 threads).  Every wave holds six accumulator tiles of its product; the two
// sums (product 0 -> out[0], product 1 -> out[1]) are rebuilt from the parity bits, XOR-ed into the
// global accumulators and, if asked, the last workgroup runs the fused finalize -- the same protocol as
// re9::tail (device-scope atomics only, no fences).
// C/D layout of the 32x32 MFMA: column n = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
// Gc[wave = (pr, h)][tile = 2 s + i][lane]: the 16 parity bits of the lane's accumulator registers of that tile
typedef uint32_t gram_parity[4][kAccTiles][64];
__device__ __forceinline__ void tail_finish(gram_parity &Gc, unsigned wave, unsigned lane, f128 *out, const fin_fuse &fz, uint64_t seq,
                                            const fin_cache *fc);

__device__ __forceinline__ void tail(const v16i (&acc)[kAccTiles], unsigned wave, unsigned lane, f128 *out, const fin_fuse &fz, uint64_t seq,
                                     const fin_cache *fc = nullptr)
{
	__shared__ gram_parity Gc;
	const unsigned b0 = 4 * (lane >> 5) + (lane & 7);
#pragma unroll
	for (int t = 0; t < kAccTiles; t++) {
		uint32_t v = 0;
#pragma unroll
		for (int r = 0; r < 16; r++)
			v |= (((uint32_t)acc[t][r] >> (b0 + (r & 3))) & 1u) << r;
		Gc[wave][t][lane] = v;
	}
	tail_finish(Gc, wave, lane, out, fz, seq, fc);
}

__device__ __forceinline__ void tail_publish(uint64_t (&z3)[2][3], f128 *out, const fin_fuse &fz, uint64_t seq, const fin_cache *fc);

// everything after the parity bits: shared by the int8 form above and the FP4 form (kernels_roundeval_fp4.hip)
__device__ __forceinline__ void tail_finish(gram_parity &Gc, unsigned wave, unsigned lane, f128 *out, const fin_fuse &fz, uint64_t seq,
                                            const fin_cache *fc)
{
	__shared__ uint64_t z3[2][3];
	__syncthreads();
	// column n' = 32 h + n of matrix (pr, s) as a GF(2^64) element (bit p = G[p][n']); z = sum_n' col * e_n'
	for (unsigned task = wave; task < 6; task += 4) {
		const unsigned pr = task / 3, s = task - 3 * pr;
		const unsigned h = lane >> 5, n = lane & 31;
		auto spread = [](uint32_t x) { return (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12); };
		const uint32_t *g0 = Gc[2 * pr + h][2 * s], *g1 = Gc[2 * pr + h][2 * s + 1];
		const uint32_t lo = spread(g0[n]) | (spread(g0[n + 32]) << 4);
		const uint32_t hi = spread(g1[n]) | (spread(g1[n + 32]) << 4);
		uint64_t z = mul_basis64((uint64_t)lo | ((uint64_t)hi << 32), lane);
#pragma unroll
		for (int mm = 32; mm >= 1; mm >>= 1)
			z ^= __shfl_xor(z, mm, 64);
		if (lane == 0) z3[pr][s] = z;
	}
	tail_publish(z3, out, fz, seq, fc);
}

// from the six GF(2^64) sums z3[product][Karatsuba term] on: the two products, their way into the global accumulators, the ticket
// and the fused finalize.  z3 is workgroup-shared and complete for the caller's own wave; the barrier in front is in here.
__device__ __forceinline__ void tail_publish(uint64_t (&z3)[2][3], f128 *out, const fin_fuse &fz, uint64_t seq, const fin_cache *fc)
{
	__shared__ f128 s_loc[2];
	const unsigned tid = threadIdx.x;
	__syncthreads();
	BN_TS(6);
	if (tid < 2)
		s_loc[tid] = kara64(z3[tid][0], z3[tid][1], z3[tid][2]);
	__syncthreads();
	BN_TS(7);
	// fc: the finalize arguments were staged in LDS at kernel entry (finalize.hpp)
	unsigned *const counter = fc ? fc->counter : fz.counter;
	f128 *const S = fc ? fc->S : fz.S;
	if (counter && gridDim.x == 1 && out == S) {
		// single workgroup: the sums never leave the chip -- finalize straight from LDS
		if (fc)
			finalize_cached(*fc, seq, s_loc);
		else
			finalize_body(fz.args, fz.S, fz.rets, fz.mail, seq, s_loc, &fz.peer);
		return;
	}
	if (tid < 4) {
		const uint64_t v = reinterpret_cast<const uint64_t *>(s_loc)[tid];
		if (v)
			atomicXor(reinterpret_cast<unsigned long long *>(out) + tid, (unsigned long long)v);
	}
	if (counter) {
		__shared__ unsigned is_last;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (tid == 0) {
			const unsigned t = atomicAdd(counter, 1u);
			is_last = (t == gridDim.x - 1) ? 1u : 0u;
		}
		__syncthreads();
		BN_TS(8);
		if (is_last) {
			if (fc)
				finalize_cached(*fc, seq);
			else
				finalize_body(fz.args, fz.S, fz.rets, fz.mail, seq, nullptr, &fz.peer);
			if (tid == 0)
				__hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

} // namespace gram
} // namespace bn
