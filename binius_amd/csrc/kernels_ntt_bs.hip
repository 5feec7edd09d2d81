// binius_amd/csrc/kernels_ntt_bs.hip -- bit-sliced additive NTT with BinaryField32b twiddles (forward;
// BASELINE config "2^24 coeffs over BinaryField32b"; semantics of crates/ntt/src/tests/reference.rs
// :68-112 with the batching of :170-204).  Data of a larger field (B64, B128) is 2 or 4 interleaved
// B32 columns -- a B32 scalar acts limb-wise -- so every shape is a batch of B32 transforms: word
// address of element p of batch (z, x) = ((z << log_y | p) << lx) | x, lx = log_x + (elem_level - 5).
//
// A butterfly is (u, v) -> (u + v*t, v + u + v*t) with a VARIABLE twiddle t per block, i.e. one
// variable x variable GF(2^32) product per butterfly.  Word-level that is ~15 table lookups + ~60
// VALU; bit-sliced (bitslice.hpp) it is ~1040 VALU per THIRTY-TWO products.
//
// Slicing.  Index p = c * S + i with c = the top 5 index bits, S = 2^(L-5).  The plane set of i is 32
// registers W[j] whose bit c is bit j of element (c, i): 32 elements that are 2^(L-5) apart.
//   * The 5 top layers (distance >= S) pair elements INSIDE a plane set; their 31 twiddles do not
//     depend on i.  They run word-level on the 32 words of the set, before the transpose, through
//     per-twiddle nibble tables in LDS (bit-sliced they would compute on half-empty bit positions).
//   * A lower layer l pairs plane sets i and i + 2^l.  The twiddle of bit position c is
//     tw(l, (c*S + i) >> (l+1)) -- and the twiddle is GF(2)-linear in its index
//     (OnTheFlyTwiddleAccess: XOR of basis values, crates/ntt/src/twiddle.rs:141-168), so its planes
//     are  pat[l][j] ^ broadcast(bit j of tw(l, i >> (l+1)) ^ coset term)  with pat a launch constant.
//
// Passes (HBM: 2 x 64 MiB each):  head = convert to plane sets + the 5 top layers;  then the lower
// layers <= 6 at a time on a 512-set tile in LDS;  tail = convert back.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "bitslice.hpp"
#include "internal.hpp"

namespace bn {

namespace {

struct ntt_bs_tables {
	// top layers: 31 launch-constant twiddles (layer b = distance 2^b among the 32 words of a set has
	// 2^(4-b) of them, slot (16 >> b) - 1 + blk... see top_slot); per twiddle 8 nibble tables of 16
	// products  tw * (e << 4p)  -- 16 KiB, staged in LDS by the head / tail kernels
	uint32_t ttab[31][8][16];
	uint32_t pat[32][32];  // [l][j]: c-dependent part of the twiddle planes of lower layer l
	uint32_t rows[32][32]; // [l][bit]: basis value of bit `bit` of the block index i >> (l+1)
	uint32_t tconst[32];   // [l]: coset contribution (uniform)
	uint32_t sub8[32];     // [l]: 1 if every twiddle of lower layer l lies in GF(2^8) (planes 8..31 of T are zero)
};

typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) volatile u4v lds_vu4; // one LDS access = one 128-bit instruction
constexpr int kSetQ = 9;     // LDS uint4 per plane set: 8 + 1 pad (bank spread)
constexpr int kTileLog = 9;  // plane sets per tile = 512

// forward: u += v*t; v += u        inverse: v += u; u += v*t   (reference.rs:96-104 / :143-151)
template <bool INV>
__device__ __forceinline__ void butterfly_planes(uint32_t (&U)[32], uint32_t (&V)[32], const uint32_t (&T)[32])
{
	uint32_t M[32];
	if (INV) {
#pragma unroll
		for (int j = 0; j < 32; j++)
			V[j] ^= U[j];
	}
	bs_mul<5>(V, T, M);
#pragma unroll
	for (int j = 0; j < 32; j++) {
		U[j] ^= M[j];
		if (!INV) V[j] ^= U[j];
	}
}

// The same butterfly when the twiddle lies in the subfield GF(2^8) = T_3 (only planes 0..7 of T can be set): a T_3
// scalar acts on the four T_3 limbs of a T_5 element separately (binary_field.rs:361-393), so the product is four
// 8-plane products (4 x 27 AND) instead of one 32-plane product (243 AND).
template <bool INV>
__device__ __forceinline__ void butterfly_planes_sub8(uint32_t (&U)[32], uint32_t (&V)[32], const uint32_t (&T)[32])
{
	uint32_t M[32];
	if (INV) {
#pragma unroll
		for (int j = 0; j < 32; j++)
			V[j] ^= U[j];
	}
#pragma unroll
	for (int i = 0; i < 4; i++)
		bs_mul<3>(V + 8 * i, T, M + 8 * i);
#pragma unroll
	for (int j = 0; j < 32; j++) {
		U[j] ^= M[j];
		if (!INV) V[j] ^= U[j];
	}
}

// slot of the twiddle of block `blk` of top layer b (b = 4: 1 block, ..., b = 0: 16 blocks)
__host__ __device__ constexpr int top_slot(int b, int blk) { return (16 >> b) - 1 + blk; }

// v * tw through the twiddle's nibble tables (LDS): 8 lookups, all lanes of a wave use the same
// twiddle, so a lookup touches one 64-byte table row (conflict-free)
// (byte offsets of the eight entries with one SDWA byte-select-and-mask each: bits 2..5 of byte k of v << 2 are the low nibble
// of byte k of v times four, of v >> 2 the high nibble times four -- 10 instead of 16 address instructions per product)
template <int B>
__device__ __forceinline__ uint32_t ntt_byte_and(uint32_t w, uint32_t m)
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
__device__ __forceinline__ uint32_t top_mul(const uint32_t *tab /*[8][16]*/, uint32_t v)
{
	const char *base = reinterpret_cast<const char *>(tab);
	const uint32_t lo = v << 2, hi = v >> 2, m = 0x3Cu;
	auto at = [&](int p, uint32_t off) { return *reinterpret_cast<const uint32_t *>(base + 64 * p + off); };
	uint32_t r = at(0, ntt_byte_and<0>(lo, m)) ^ at(1, ntt_byte_and<0>(hi, m));
	r ^= at(2, ntt_byte_and<1>(lo, m)) ^ at(3, ntt_byte_and<1>(hi, m));
	r ^= at(4, ntt_byte_and<2>(lo, m)) ^ at(5, ntt_byte_and<2>(hi, m));
	r ^= at(6, ntt_byte_and<3>(lo, m)) ^ at(7, ntt_byte_and<3>(hi, m));
	return r;
}

// The five top layers act among the 32 words W[c] = element (c, i) of one plane set BEFORE it is
// transposed (forward) / after it is transposed back (inverse): butterflies (c, c + 2^b) with the
// launch-constant twiddle of block c >> (b+1).  Word-level with nibble tables this is ~30 VALU per
// butterfly; bit-sliced inside the registers it would run on half-empty bit positions (~78).
// skip_rounds drops the highest layers: only b < n_top are applied.
template <bool INV>
__device__ __forceinline__ void top_layers(uint32_t (&W)[32], const uint32_t *ttab_lds, uint32_t n_top)
{
#pragma unroll
	for (int bb = 0; bb < 5; bb++) {
		const int b = INV ? bb : 4 - bb;
		if ((uint32_t)b >= n_top) continue;
#pragma unroll
		for (int c = 0; c < 32; c++) {
			if ((c >> b) & 1) continue;
			const uint32_t *tab = ttab_lds + top_slot(b, c >> (b + 1)) * 128;
			uint32_t u = W[c], v = W[c + (1 << b)];
			if (INV) {
				v ^= u;
				u ^= top_mul(tab, v);
			} else {
				u ^= top_mul(tab, v);
				v ^= u;
			}
			W[c] = u;
			W[c + (1 << b)] = v;
		}
	}
}

__device__ __forceinline__ void stage_top_tables(uint32_t *lds, const ntt_bs_tables *tb)
{
	const uint32_t *src = &tb->ttab[0][0][0];
	for (unsigned q = threadIdx.x; q < 31 * 128; q += blockDim.x)
		lds[q] = src[q];
	__syncthreads();
}

// ---- head: standard layout -> plane sets; forward: then the five in-register layers (16, 8, .., 1)
// batch beta = z * 2^lx + x with z = blockIdx.y
template <bool INV>
__global__ __launch_bounds__(256) void k_ntt_bs_head(const uint32_t *__restrict__ data, uint4 *__restrict__ bs, uint64_t S, uint32_t lx,
                                                     uint32_t log_y, const ntt_bs_tables *__restrict__ tb, uint32_t n_top)
{
	__shared__ uint32_t ttab_lds[31 * 128];
	if (!INV) stage_top_tables(ttab_lds, tb);
	// the interleaved transforms x are the fastest index over the lanes: the 2^lx words of one (c, i)
	// are contiguous in memory, so the loads coalesce whatever lx is
	const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	const uint64_t i = g >> lx, x = g & (((uint64_t)1 << lx) - 1);
	if (i >= S) return;
	const uint64_t beta = ((uint64_t)blockIdx.y << lx) | x;
	data += ((uint64_t)blockIdx.y << (log_y + lx)) + x;
	bs += beta * S * 8;
	uint32_t W[32];
#pragma unroll
	for (int c = 0; c < 32; c++)
		W[c] = data[((uint64_t)c * S + i) << lx];
	if (!INV) top_layers<false>(W, ttab_lds, n_top);
	transpose32(W);
	uint4 *dst = bs + i * 8;
#pragma unroll
	for (int k = 0; k < 8; k++)
		dst[k] = uint4{W[4 * k], W[4 * k + 1], W[4 * k + 2], W[4 * k + 3]};
}

// ---- tail: (inverse: the five in-register layers 1, 2, .., 16, then) plane sets -> standard layout
template <bool INV>
__global__ __launch_bounds__(256) void k_ntt_bs_tail(const uint4 *__restrict__ bs, uint32_t *__restrict__ data, uint64_t S, uint32_t lx,
                                                     uint32_t log_y, const ntt_bs_tables *__restrict__ tb, uint32_t n_top)
{
	__shared__ uint32_t ttab_lds[31 * 128];
	if (INV) stage_top_tables(ttab_lds, tb);
	const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; // x fastest, as in the head
	const uint64_t i = g >> lx, x = g & (((uint64_t)1 << lx) - 1);
	if (i >= S) return;
	const uint64_t beta = ((uint64_t)blockIdx.y << lx) | x;
	data += ((uint64_t)blockIdx.y << (log_y + lx)) + x;
	bs += beta * S * 8;
	uint32_t W[32];
	const uint4 *src = bs + i * 8;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const uint4 v = src[k];
		W[4 * k] = v.x;
		W[4 * k + 1] = v.y;
		W[4 * k + 2] = v.z;
		W[4 * k + 3] = v.w;
	}
	transpose32(W); // (an involution)
	if (INV) top_layers<true>(W, ttab_lds, n_top);
#pragma unroll
	for (int c = 0; c < 32; c++)
		data[((uint64_t)c * S + i) << lx] = W[c];
}

// ---- R lower layers l_lo + R - 1 .. l_lo on a tile of 512 plane sets held in LDS.
// Tile-local set number s (9 bits) <-> index bits: run A = bits [0, n_lo), run B = bits
// [l_lo, l_lo + 9 - n_lo) with n_lo = min(9 - R, l_lo); the butterfly bit of layer l_lo + t is local
// bit n_lo + t.  The other index bits enumerate the tiles.
// CONV (only with l_lo == 0, i.e. a tile of 512 CONSECUTIVE sets): the conversion between the
// standard layout and the plane sets rides in this pass -- forward: the last pass stores elements
// (no tail kernel), inverse: the first pass loads elements (no head kernel); `data`, lx, log_y as in
// the head / tail kernels.
template <bool INV, bool CONV>
__global__ __launch_bounds__(256, 2) void k_ntt_bs_pass(uint4 *__restrict__ bs, uint64_t S, uint32_t l_lo, uint32_t R, uint32_t n_lo,
                                                        const ntt_bs_tables *__restrict__ tb, uint32_t *__restrict__ data, uint32_t lx,
                                                        uint32_t log_y, uint32_t wg_bar)
{
	extern __shared__ __attribute__((aligned(16))) uint4 tile[]; // [512][kSetQ]
	lds_vu4 *tile3 = (lds_vu4 *)(__attribute__((address_space(3))) void *)tile;
	const unsigned tid = threadIdx.x;
	bs += (uint64_t)blockIdx.y * S * 8; // batch
	if (CONV) {
		const uint64_t beta = blockIdx.y;
		data += ((beta >> lx) << (log_y + lx)) + (beta & (((uint64_t)1 << lx) - 1));
	}
	const uint32_t gap = l_lo - n_lo; // tile bits between the two local runs
	const uint64_t b = blockIdx.x;
	const uint64_t i_tile = ((b & (((uint64_t)1 << gap) - 1)) << n_lo) | ((b >> gap) << (l_lo + kTileLog - n_lo));
	auto index_of = [&](unsigned s) -> uint64_t { return i_tile | (s & ((1u << n_lo) - 1)) | ((uint64_t)(s >> n_lo) << l_lo); };
	if (CONV && INV) {
		// elements -> plane sets, two sets per thread (consecutive i across the lanes: coalesced words)
#pragma unroll 1
		for (unsigned h = 0; h < 2; h++) {
			const unsigned sset = tid + 256 * h;
			const uint64_t i = index_of(sset);
			uint32_t W[32];
#pragma unroll
			for (int c = 0; c < 32; c++)
				W[c] = data[((uint64_t)c * S + i) << lx];
			transpose32(W);
#pragma unroll
			for (int k = 0; k < 8; k++)
				tile[sset * kSetQ + k] = uint4{W[4 * k], W[4 * k + 1], W[4 * k + 2], W[4 * k + 3]};
		}
	} else {
		// load: 512 sets x 8 chunks of 16 B
#pragma unroll 4
		for (unsigned k = 0; k < 16; k++) {
			const unsigned idx = tid + 256 * k, s = idx >> 3, ch = idx & 7;
			tile[s * kSetQ + ch] = bs[index_of(s) * 8 + ch];
		}
	}
	__syncthreads();
	for (int tt = 0; tt < (int)R; tt++) {
		const int t = INV ? tt : (int)R - 1 - tt; // forward: high layer first
		const uint32_t l = l_lo + (uint32_t)t;
		const unsigned pos = n_lo + (unsigned)t;
		// butterfly tid: u set = tid with a zero inserted at bit `pos`
		const unsigned s_u = ((tid >> pos) << (pos + 1)) | (tid & ((1u << pos) - 1));
		const unsigned s_v = s_u | (1u << pos);
		uint32_t U[32], V[32], T[32];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			// (volatile: keeps each access one ds_read_b128 / ds_write_b128 -- left alone the compiler re-splits
			// them into ds_read2_b32 / ds_write2_b32, whose 32-bank mapping makes the 144-byte set stride a
			// 4-way conflict: 74 % of this kernel's LDS cycles were conflict cycles)
			const u4v a = tile3[s_u * kSetQ + k], c = tile3[s_v * kSetQ + k];
			U[4 * k] = a.x;
			U[4 * k + 1] = a.y;
			U[4 * k + 2] = a.z;
			U[4 * k + 3] = a.w;
			V[4 * k] = c.x;
			V[4 * k + 1] = c.y;
			V[4 * k + 2] = c.z;
			V[4 * k + 3] = c.w;
		}
		// twiddle of bit position c: pat (c part) ^ tbase (i part + coset part), all XOR-linear.  The
		// index bits of this set split into the tile's (wave-uniform: scalar loop) and the <= 9 local
		// ones (one conditional XOR each, basis values through scalar loads) -- no divergent loop.
		uint32_t tbase = tb->tconst[l];
		{
			uint64_t qt = i_tile >> (l + 1);
			for (unsigned bit = 0; qt; bit++, qt >>= 1)
				if (qt & 1) tbase ^= tb->rows[l][bit];
#pragma unroll
			for (unsigned k = 0; k < (unsigned)kTileLog; k++) {
				const unsigned gp = k < n_lo ? k : l_lo + k - n_lo; // index bit of local bit k
				if (gp > l) {
					const uint32_t rv = tb->rows[l][gp - (l + 1)];
					tbase ^= ((s_u >> k) & 1) ? rv : 0u;
				}
			}
		}
#pragma unroll
		for (int j = 0; j < 32; j++)
			T[j] = tb->pat[l][j] ^ (uint32_t)__builtin_amdgcn_sbfe((int)tbase, j, 1);
		if (tb->sub8[l]) // (uniform)
			butterfly_planes_sub8<INV>(U, V, T);
		else
			butterfly_planes<INV>(U, V, T);
#pragma unroll
		for (int k = 0; k < 8; k++) {
			tile3[s_u * kSetQ + k] = u4v{U[4 * k], U[4 * k + 1], U[4 * k + 2], U[4 * k + 3]};
			tile3[s_v * kSetQ + k] = u4v{V[4 * k], V[4 * k + 1], V[4 * k + 2], V[4 * k + 3]};
		}
		// Who reads what this layer wrote?  The thread that handles set s at butterfly bit `pos` is s with that bit removed,
		// so for pos <= 6 its wave is s >> 7 whatever pos is: between two layers whose butterfly bits are both <= 6 every set
		// stays inside one wave, and the wave's own LDS operations are performed in order -- no workgroup barrier (the waves of
		// a tile drift apart instead of waiting for the slowest after every layer).  BN_NTT_WG_BARRIERS=1: round 3's form.
		const int tn = tt + 1 < (int)R ? (INV ? tt + 1 : (int)R - 2 - tt) : -1;
		const unsigned pos_next = tn >= 0 ? n_lo + (unsigned)tn : 99u;
		if (!wg_bar && pos <= 6 && pos_next <= 6) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		} else {
			__syncthreads();
		}
	}
	if (CONV && !INV) {
		// plane sets -> elements
#pragma unroll 1
		for (unsigned h = 0; h < 2; h++) {
			const unsigned sset = tid + 256 * h;
			const uint64_t i = index_of(sset);
			uint32_t W[32];
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const uint4 v = tile[sset * kSetQ + k];
				W[4 * k] = v.x;
				W[4 * k + 1] = v.y;
				W[4 * k + 2] = v.z;
				W[4 * k + 3] = v.w;
			}
			transpose32(W);
#pragma unroll
			for (int c = 0; c < 32; c++)
				data[((uint64_t)c * S + i) << lx] = W[c];
		}
		return;
	}
#pragma unroll 4
	for (unsigned k = 0; k < 16; k++) {
		const unsigned idx = tid + 256 * k, s = idx >> 3, ch = idx & 7;
		bs[index_of(s) * 8 + ch] = tile[s * kSetQ + ch];
	}
}

// ---- the same R <= 7 lower layers with the plane sets held in REGISTERS (round 4) ----------------------------------------
// k_ntt_bs_pass above moves every plane set through LDS once per layer (16 + 16 ds_*_b128 per thread and layer, all four
// waves of a tile in their LDS phase at the same time, then all in their product phase): VALU issue 59 % busy.  Here a WAVE
// owns 128 plane sets for the whole pass, two per lane (X0, X1: 64 registers), and a layer is the in-lane butterfly (X0, X1):
// no LDS, no barrier, nothing shared between waves.  Between two layers ONE plane set per lane changes hands:
// local index bits live either in the slot (X0 / X1) or in one of six lane bits; the layer about to run needs its bit in the
// slot, so the lanes that differ in that lane bit swap X1 (side 0) against X0 (side 1) -- side 0 then holds (its X0, the
// partner's X0), side 1 (the partner's X1, its X1): both are butterfly pairs of the new bit, and the bit that was in the slot
// now lives in the lane bit.  Every bit is a butterfly bit at most once, so the permutation is never undone; it is tracked in
// a packed scalar (phys) and resolved when the sets are stored.  The six lane bits are chosen so that each swap is one or two
// instructions per register: lane ^ 32 and lane ^ 16 are v_permlane32_swap / v_permlane16_swap (gfx950), row_mirror and
// row_half_mirror are DPP moves whose two sides are whole banks (bank_mask), lane ^ 2 and lane ^ 1 DPP moves + selects.
// Encoded lane bits (each operation flips exactly one of them): e5 = j5, e4 = j4, e3 = j3, e2 = j2 ^ j3, e1 = j1 ^ j2,
// e0 = j0 ^ j2 for lane j.
constexpr int kRegLog = 7; // plane sets per wave = 128

template <int B>
__device__ __forceinline__ void reg_swap(uint32_t (&X0)[32], uint32_t (&X1)[32], unsigned side)
{
#pragma unroll
	for (int j = 0; j < 32; j++) {
		if constexpr (B == 5) {
			const auto r = __builtin_amdgcn_permlane32_swap(X0[j], X1[j], false, false); // X0[upper half] <-> X1[lower half]
			X0[j] = r[0];
			X1[j] = r[1];
		} else if constexpr (B == 4) {
			const auto r = __builtin_amdgcn_permlane16_swap(X0[j], X1[j], false, false); // X0[odd rows] <-> X1[even rows]
			X0[j] = r[0];
			X1[j] = r[1];
		} else if constexpr (B == 3) {
			// row_mirror (lane ^ 15 inside a row of 16): side = j3 = banks 2, 3
			const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp((int)X1[j], (int)X0[j], 0x140, 0xF, 0x3, false);
			const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp((int)X0[j], (int)X1[j], 0x140, 0xF, 0xC, false);
			X0[j] = n0;
			X1[j] = n1;
		} else if constexpr (B == 2) {
			// row_half_mirror (lane ^ 7 inside a group of 8): side = j2 ^ j3 = banks 1, 2
			const uint32_t n1 = (uint32_t)__builtin_amdgcn_update_dpp((int)X1[j], (int)X0[j], 0x141, 0xF, 0x9, false);
			const uint32_t n0 = (uint32_t)__builtin_amdgcn_update_dpp((int)X0[j], (int)X1[j], 0x141, 0xF, 0x6, false);
			X0[j] = n0;
			X1[j] = n1;
		} else {
			// inside a quad: lane ^ 2 (B == 1) or lane ^ 1 (B == 0); the sides are not banks -- select
			constexpr int ctrl = B == 1 ? 0x4E : 0xB1;
			const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)X0[j], ctrl, 0xF, 0xF, true);
			const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)X1[j], ctrl, 0xF, 0xF, true);
			X0[j] = side ? r1 : X0[j];
			X1[j] = side ? X1[j] : r0;
		}
	}
}

template <bool INV, bool CONV>
__global__ __launch_bounds__(256, 2) void k_ntt_bs_pass_reg(uint4 *__restrict__ bs, uint64_t S, uint32_t l_lo, uint32_t R, uint32_t n_lo,
                                                            const ntt_bs_tables *__restrict__ tb, uint32_t *__restrict__ data, uint32_t lx,
                                                            uint32_t log_y)
{
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	bs += (uint64_t)blockIdx.y * S * 8; // batch
	if (CONV) {
		const uint64_t beta = blockIdx.y;
		data += ((beta >> lx) << (log_y + lx)) + (beta & (((uint64_t)1 << lx) - 1));
	}
	const uint32_t gap = l_lo - n_lo; // block bits between the two local runs
	const uint64_t b = (uint64_t)blockIdx.x * 4 + wave;
	const uint64_t i_tile = ((b & (((uint64_t)1 << gap) - 1)) << n_lo) | ((b >> gap) << (l_lo + kRegLog - n_lo));
	auto index_of = [&](unsigned s) -> uint64_t { return i_tile | (s & ((1u << n_lo) - 1)) | ((uint64_t)(s >> n_lo) << l_lo); };
	// encoded lane bits
	const unsigned j2 = (lane >> 2) & 1, j3 = (lane >> 3) & 1;
	const unsigned E = (lane & 0x30) | (j3 << 3) | ((j2 ^ j3) << 2) | ((((lane >> 1) & 1) ^ j2) << 1) | ((lane & 1) ^ j2);
	// phys: 4 bits per local index bit k = the lane bit that holds it (15: the slot).  The first butterfly bit starts in the
	// slot, the other six local bits take the lane bits in increasing order.
	const unsigned k_first = INV ? n_lo : n_lo + R - 1;
	uint32_t phys = 0;
	{
		unsigned nb = 0;
		for (unsigned k = 0; k < (unsigned)kRegLog; k++)
			phys |= (k == k_first ? 15u : nb++) << (4 * k);
	}
	auto local_index = [&](uint32_t ph, unsigned slot) -> unsigned { // local set number of this lane's X_slot
		unsigned s = 0;
#pragma unroll
		for (unsigned k = 0; k < (unsigned)kRegLog; k++) {
			const unsigned pb = (ph >> (4 * k)) & 15u;
			s |= (pb == 15u ? slot : ((E >> pb) & 1u)) << k;
		}
		return s;
	};
	uint32_t X0[32], X1[32];
	if (CONV && INV) {
		// elements -> plane sets (the words of one set are S elements apart; the lanes of a wave cover 128 consecutive i)
#pragma unroll 1
		for (unsigned h = 0; h < 2; h++) {
			const uint64_t i = index_of(local_index(phys, h));
			uint32_t W[32];
#pragma unroll
			for (int c = 0; c < 32; c++)
				W[c] = data[((uint64_t)c * S + i) << lx];
			transpose32(W);
#pragma unroll
			for (int c = 0; c < 32; c++) {
				if (h == 0) X0[c] = W[c];
				else X1[c] = W[c];
			}
		}
	} else {
		const uint4 *p0 = bs + index_of(local_index(phys, 0)) * 8, *p1 = bs + index_of(local_index(phys, 1)) * 8;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint4 a = p0[k], c = p1[k];
			X0[4 * k] = a.x; X0[4 * k + 1] = a.y; X0[4 * k + 2] = a.z; X0[4 * k + 3] = a.w;
			X1[4 * k] = c.x; X1[4 * k + 1] = c.y; X1[4 * k + 2] = c.z; X1[4 * k + 3] = c.w;
		}
	}
	unsigned k_prev = k_first;
#pragma unroll 1
	for (unsigned tt = 0; tt < R; tt++) {
		const unsigned k = INV ? n_lo + tt : n_lo + R - 1 - tt; // local index bit of this layer (forward: high layer first)
		const uint32_t l = l_lo + (k - n_lo);
		if (tt) {
			const unsigned pb = __builtin_amdgcn_readfirstlane((phys >> (4 * k)) & 15u);
			const unsigned side = (E >> pb) & 1u;
			switch (pb) { // (uniform)
			case 0: reg_swap<0>(X0, X1, side); break;
			case 1: reg_swap<1>(X0, X1, side); break;
			case 2: reg_swap<2>(X0, X1, side); break;
			case 3: reg_swap<3>(X0, X1, side); break;
			case 4: reg_swap<4>(X0, X1, side); break;
			default: reg_swap<5>(X0, X1, side); break;
			}
			phys = (phys & ~((15u << (4 * k)) | (15u << (4 * k_prev)))) | (15u << (4 * k)) | (pb << (4 * k_prev));
			k_prev = k;
		}
		// twiddle of bit position c: pat (c part) ^ tbase (i part + coset part), all XOR-linear; this lane's u set is X0, whose
		// index has bit l clear; the bits above l come from the block (uniform: scalar loop) and from the lane bits
		uint32_t tbase = tb->tconst[l];
		{
			uint64_t qt = i_tile >> (l + 1);
			for (unsigned bit = 0; qt; bit++, qt >>= 1)
				if (qt & 1) tbase ^= tb->rows[l][bit];
#pragma unroll
			for (unsigned kk = 0; kk < (unsigned)kRegLog; kk++) {
				const unsigned gp = kk < n_lo ? kk : l_lo + kk - n_lo; // index bit of local bit kk
				if (gp > l) { // (uniform; never the slot bit, which is bit l itself)
					const uint32_t rv = tb->rows[l][gp - (l + 1)];
					const unsigned pb = (phys >> (4 * kk)) & 15u;
					tbase ^= ((E >> pb) & 1u) ? rv : 0u;
				}
			}
		}
		uint32_t T[32];
#pragma unroll
		for (int j = 0; j < 32; j++)
			T[j] = tb->pat[l][j] ^ (uint32_t)__builtin_amdgcn_sbfe((int)tbase, j, 1);
		if (tb->sub8[l]) // (uniform)
			butterfly_planes_sub8<INV>(X0, X1, T);
		else
			butterfly_planes<INV>(X0, X1, T);
	}
	if (CONV && !INV) {
		// plane sets -> elements
#pragma unroll 1
		for (unsigned h = 0; h < 2; h++) {
			const uint64_t i = index_of(local_index(phys, h));
			uint32_t W[32];
#pragma unroll
			for (int c = 0; c < 32; c++)
				W[c] = h == 0 ? X0[c] : X1[c];
			transpose32(W);
#pragma unroll
			for (int c = 0; c < 32; c++)
				data[((uint64_t)c * S + i) << lx] = W[c];
		}
		return;
	}
	uint4 *q0 = bs + index_of(local_index(phys, 0)) * 8, *q1 = bs + index_of(local_index(phys, 1)) * 8;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		q0[k] = uint4{X0[4 * k], X0[4 * k + 1], X0[4 * k + 2], X0[4 * k + 3]};
		q1[k] = uint4{X1[4 * k], X1[4 * k + 1], X1[4 * k + 2], X1[4 * k + 3]};
	}
}

// OnTheFlyTwiddleAccess::get (twiddle.rs:141-168): XOR of the layer's basis values over the index bits
uint32_t host_twiddle(const uint64_t *s_evals, uint32_t log_domain, uint32_t layer, uint64_t index)
{
	const uint64_t *row = s_evals + (size_t)layer * BN_NTT_MAX_DIM;
	const int n_bits = (int)log_domain - 1 - (int)layer;
	uint64_t t = 0;
	for (int b = 0; b < n_bits; b++)
		if ((index >> b) & 1) t ^= row[b];
	return (uint32_t)t;
}

} // namespace

// Forward / inverse NTT of 2^(lx + log_z) interleaved B32 transforms of 2^log_y elements each; skip_rounds drops the highest layers.
// d_scratch: ntt_bs_scratch_bytes() bytes, 256-byte aligned.
// Returns hipErrorNotSupported for shapes this path does not cover (the caller falls back).
size_t ntt_bs_scratch_bytes(uint32_t log_words) { return ((size_t)4 << log_words) + 256; }

template <bool INV>
static hipError_t run_ntt_bs(hipStream_t s, void *data, const uint64_t *h_s_evals, uint32_t log_domain, uint32_t lx, uint32_t log_y,
                             uint32_t log_z, uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds, void *d_scratch, ntt_bs_cache *cache)
{
	const uint32_t L = log_y;
	if (L < 5 + kTileLog || L > 31 || lx + log_z > 12) return hipErrorNotSupported;
	const unsigned n_batch = 1u << (lx + log_z);
	const uint32_t NB = L - 5;           // lower layers
	const uint64_t S = (uint64_t)1 << NB; // plane sets per transform
	const uint32_t base = log_domain - (log_y + coset_bits);
	// ---- launch constants: rebuilt and uploaded only when the domain / coset / size changed since the
	// previous call of this context (a prover transforms many columns over the same domain)
	uint64_t key = 0xcbf29ce484222325ull; // FNV-1a over everything the tables depend on
	auto mix = [&](uint64_t v) {
		for (int k = 0; k < 8; k++) {
			key ^= (v >> (8 * k)) & 0xFF;
			key *= 0x100000001b3ull;
		}
	};
	mix(L);
	mix(log_domain);
	mix(coset);
	mix(coset_bits);
	for (uint32_t l = 0; l < L; l++)
		for (uint32_t b = 0; b + 1 + base + l < log_domain; b++) mix(h_s_evals[(size_t)(base + l) * BN_NTT_MAX_DIM + b]);
	ntt_bs_tables *d_tb = (ntt_bs_tables *)cache->d_tables;
	hipError_t e = hipSuccess;
	if (!cache->valid || cache->key != key) {
	static thread_local ntt_bs_tables tb;
	std::memset(&tb, 0, sizeof(tb));
	for (uint32_t b = 0; b < 5; b++) {
		const uint32_t l = NB + b;
		for (uint32_t blk = 0; blk < (16u >> b); blk++) {
			const uint32_t tw = host_twiddle(h_s_evals, log_domain, base + l, (coset << (L - 1 - l)) | blk);
			for (uint32_t p = 0; p < 8; p++)
				for (uint32_t e = 0; e < 16; e++)
					tb.ttab[top_slot((int)b, (int)blk)][p][e] = (uint32_t)mul_slow(f128{tw, 0}, f128{(uint64_t)e << (4 * p), 0}).lo;
		}
	}
	for (uint32_t l = 0; l < NB; l++) {
		// (c*S + i) >> (l+1) = (c << (NB - l - 1)) + (i >> (l+1)): no carries, the twiddle splits
		for (uint32_t c = 0; c < 32; c++) {
			const uint32_t tw = host_twiddle(h_s_evals, log_domain, base + l, (uint64_t)c << (NB - l - 1));
			for (uint32_t j = 0; j < 32; j++)
				if ((tw >> j) & 1) tb.pat[l][j] |= 1u << c;
		}
		for (uint32_t bit = 0; bit < NB - l - 1 && bit < 32; bit++)
			tb.rows[l][bit] = host_twiddle(h_s_evals, log_domain, base + l, (uint64_t)1 << bit);
		tb.tconst[l] = host_twiddle(h_s_evals, log_domain, base + l, coset << (L - 1 - l));
		uint32_t any = tb.tconst[l];
		for (uint32_t c = 0; c < 32; c++) any |= host_twiddle(h_s_evals, log_domain, base + l, (uint64_t)c << (NB - l - 1));
		for (uint32_t bit = 0; bit < 32; bit++) any |= tb.rows[l][bit];
		static const bool no_sub8 = bn::settled_knob("BN_NTT_NO_SUB8") != nullptr;
		tb.sub8[l] = (any < 256 && !no_sub8) ? 1u : 0u;
	}
	e = hipMemcpyAsync(d_tb, &tb, sizeof(tb), hipMemcpyHostToDevice, s);
	if (e != hipSuccess) return e;
	e = hipStreamSynchronize(s); // tb is reused by the next rebuild
	if (e != hipSuccess) return e;
	cache->valid = true;
	cache->key = key;
	}
	uint4 *bs = (uint4 *)d_scratch;

	// layers log_y - skip_rounds - 1 .. 0 are applied (reference.rs:88): of the five in-register layers
	// the lowest n_top, of the NB lower layers the lowest n_low
	const uint32_t n_top = skip_rounds >= 5 ? 0 : 5 - skip_rounds;
	const uint32_t n_low = skip_rounds > 5 ? NB - (skip_rounds - 5) : NB;
	const size_t lds = (size_t)(1 << kTileLog) * kSetQ * sizeof(uint4);
	e = func_lds_limit(reinterpret_cast<const void *>(&k_ntt_bs_pass<INV, false>), (int)lds);
	if (e == hipSuccess) e = func_lds_limit(reinterpret_cast<const void *>(&k_ntt_bs_pass<INV, true>), (int)lds);
	if (e != hipSuccess) return e;
	// lower layers, at most 7 per pass: forward from NB-1 down to 0, inverse from 0 up to NB-1
	std::vector<std::pair<uint32_t, uint32_t>> plan; // (l_lo, R), highest layers first
	for (uint32_t hi = n_low; hi > 0;) {
		const uint32_t n_pass = (hi + 6) / 7;
		const uint32_t R = (hi + n_pass - 1) / n_pass; // even split
		plan.push_back({hi - R, R});
		hi -= R;
	}
	// the pass over layers R-1..0 works on consecutive sets and converts the layout itself: forward it
	// is the last kernel (no tail), inverse the first (no head)
	// -- unless the transforms are interleaved (lx > kMergeMaxLx): the merged conversion walks one transform,
	// i.e. words 2^lx apart, and every 4-byte access costs a whole sector; head and tail put x across the lanes
	static const uint32_t merge_max_lx = [] {
		const char *v = bn::settled_knob("BN_NTT_MERGE_MAX_LX");
		return v ? (uint32_t)atoi(v) : 1u;
	}();
	const bool merged = !plan.empty() && lx <= merge_max_lx;
	const char *wb = bn::settled_knob("BN_NTT_WG_BARRIERS"); // (read per call: a measurement knob)
	const uint32_t wg_bar = wb && wb[0] == '1' ? 1u : 0u;
	const dim3 ht_grid((unsigned)(((S << lx) + 255) / 256), 1u << log_z);
	if (!(INV && merged))
		hipLaunchKernelGGL(k_ntt_bs_head<INV>, ht_grid, dim3(256), 0, s, (const uint32_t *)data, bs, S, lx, log_y, d_tb, n_top);
	static const bool reg_pass = [] {
		const char *v = bn::settled_knob("BN_NTT_REG_PASS"); // 0: the LDS-tile passes of rounds 1 - 3
		return !(v && v[0] == '0');
	}();
	for (size_t k = 0; k < plan.size(); k++) {
		const auto &pr = INV ? plan[plan.size() - 1 - k] : plan[k];
		const uint32_t l_lo = pr.first, R = pr.second;
		if (reg_pass) {
			// a wave owns 128 plane sets (kRegLog local bits), four waves per workgroup: the same grid as the LDS tiles
			const uint32_t Qr = kRegLog - R, n_lo_r = Qr < l_lo ? Qr : l_lo;
			// (one block per wave, 162 registers: three waves per SIMD.  A persistent grid of two waves per SIMD walking two blocks
			// each -- no fourth wave running alone at the end -- was measured slower: 0.274 - 0.286 against 0.252 - 0.264 ms)
			const dim3 grid_r((unsigned)(S >> (kRegLog + 2)), n_batch);
			if (l_lo == 0 && merged)
				hipLaunchKernelGGL((k_ntt_bs_pass_reg<INV, true>), grid_r, dim3(256), 0, s, bs, S, l_lo, R, n_lo_r, d_tb, (uint32_t *)data, lx, log_y);
			else
				hipLaunchKernelGGL((k_ntt_bs_pass_reg<INV, false>), grid_r, dim3(256), 0, s, bs, S, l_lo, R, n_lo_r, d_tb, (uint32_t *)data, lx, log_y);
			continue;
		}
		const uint32_t Q = kTileLog - R;
		const uint32_t n_lo = Q < l_lo ? Q : l_lo;
		const dim3 grid((unsigned)(S >> kTileLog), n_batch);
		if (l_lo == 0 && merged)
			hipLaunchKernelGGL((k_ntt_bs_pass<INV, true>), grid, dim3(256), lds, s, bs, S, l_lo, R, n_lo, d_tb, (uint32_t *)data, lx, log_y, wg_bar);
		else
			hipLaunchKernelGGL((k_ntt_bs_pass<INV, false>), grid, dim3(256), lds, s, bs, S, l_lo, R, n_lo, d_tb, (uint32_t *)data, lx, log_y, wg_bar);
	}
	if (INV || !merged)
		hipLaunchKernelGGL(k_ntt_bs_tail<INV>, ht_grid, dim3(256), 0, s, bs, (uint32_t *)data, S, lx, log_y, d_tb, n_top);
	return hipGetLastError();
}

hipError_t launch_ntt_bs(hipStream_t s, bool inverse, void *data, const uint64_t *h_s_evals, uint32_t log_domain, uint32_t lx,
                         uint32_t log_y, uint32_t log_z, uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds, void *d_scratch,
                         ntt_bs_cache *cache)
{
	return inverse ? run_ntt_bs<true>(s, data, h_s_evals, log_domain, lx, log_y, log_z, coset, coset_bits, skip_rounds, d_scratch, cache)
	               : run_ntt_bs<false>(s, data, h_s_evals, log_domain, lx, log_y, log_z, coset, coset_bits, skip_rounds, d_scratch, cache);
}

size_t ntt_bs_tables_bytes() { return sizeof(ntt_bs_tables); }

} // namespace bn
