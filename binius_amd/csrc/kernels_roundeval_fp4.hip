// binius_amd/csrc/kernels_roundeval_fp4.hip -- round evaluation of the bivariate product (round 0: no fold before it)
// with the GF(2) Gram products of gram.hpp on the FP4 matrix path: v_mfma_scale_f32_32x32x64_f8f6f4 with both operands
// E2M1 runs K = 64 in the time the int8 instruction takes for K = 32 (tools/fp4_probe.hip: 14.6 ns against 15.4 ns per
// instruction and SIMD), and a register holds eight K-entries instead of four, so the operand masks per point halve
// as well.  Same mathematics as kernels_roundeval_mfma.hip (one Karatsuba level, six accumulator tiles per wave, the
// tail of gram.hpp); what changes is the operand encoding:
//
//  * An operand entry is a NIBBLE of the data with one bit kept: code 0001 = 0.5, 0010 = 1, 0100 = 2 (E2M1; the probe
//    confirms the subnormal).  Bit 3 of a nibble is the sign bit of the format -- 1000 decodes to -0 -- so the staging
//    also writes, per pair of limbs, a word W3 that carries bit 3 of limb 2q at position 2 and bit 3 of limb 2q + 1 at
//    position 1; the rows / columns that belong to bit 3 read W3 instead of the data word (a per-lane LDS offset, no
//    extra instruction) and mask it like everybody else.
//  * Products are 2^(e_row + e_col), the f32 accumulator of an entry holds count * 2^(e_row + e_col) exactly (count <
//    2^24: a workgroup sees at most 2^22 points), and the parity wanted is bit 0 of the count.
//  * K order: operand position (lane half, register, nibble) of A meets the same position of B (probe), so any
//    arrangement of the 64 points of a k-step works as long as u and v use the same one.
//
// LDS tile of 256 points, 24 KiB: T4[set 0..3][limb 0..3][k-step 0..3][nibble index 0..7][8 words], a word = that nibble
// of the limb for eight points; W3[set][limb pair 0..1][k-step][nibble index][8 words]; inside a 64-word block the 16-byte
// chunk of (nibble index c, k half) sits at chunk c + 8 * khalf (conflict-free for the staging writes and the operand
// reads alike, PMC-checked).  A reader lane (row i, k half) takes the chunk of nibble index i >> 2: lanes broadcast.
// Staging: lane = point.  The 8 x 8 nibble transpose across eight lanes is three exchanges -- 16-bit halves with lane
// 7 - j (DPP row_half_mirror), bytes with lane j ^ 1, nibbles with lane j ^ 2 (DPP quad_perm) -- the upper four lanes
// taking their decisions from the mirrored index so that every lane ends up with the same point order.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "gram_fp4.hpp"

namespace bn {

using namespace gram;

using namespace gram4;

// Element loads go straight into LDS (global_load_lds_dwordx4: no registers held while they fly), ONE tile ahead: with
// K = 64 the Gram k-steps of a tile take ~0.5 us, a fraction of what a loaded HBM takes to answer, and a second register
// set for a deeper register prefetch does not survive next to 96 accumulator registers (two loop bodies: 101 - 191
// spilled registers in three shapes).  One 16 KiB slot of raw elements (read into registers right after a tile's k-steps
// and refilled at once) + the 24 KiB operand tile per workgroup = 47 KiB and 154 registers: THREE workgroups per CU
// (__launch_bounds__(256, 3), grid cap = 3 per CU); per tile two barriers: raw -> operand tile (VALU), then the Gram
// k-steps (matrix pipe) -- the workgroups of a CU interleave the two phases.  (Three raw slots at two workgroups per CU
// were measured slower: DESIGN.md 4.5b.)
constexpr int kRawSlots = 1;
#ifdef BN_FP4_PHASES
__device__ unsigned long long fp4_phase_cycles[8];
#define FP4_CLK(i)                                                 \
	do {                                                           \
		const unsigned long long now_ = clock64();                 \
		if (threadIdx.x == 0 && blockIdx.x == 0) fp4_phase_cycles[i] += now_ - last_; \
		last_ = now_;                                              \
	} while (0)
#else
#define FP4_CLK(i)
#endif
// (This is the three-workgroups-per-CU form: ragged sizes and launches below two tiles per CU.  Whole tiles from 2^20 points on run
// k_roundeval_fp4_ws further down: stager waves and software-pipelined Gram waves.)
// MIX: the second pair of sets holds hi ^ lo (evaluation at infinity); otherwise lo itself (two independent products: the
// two halves of an inner product).
// NT: the LDS-DMA loads carry the non-temporal hint (HBM-resident sizes: every element is read once per launch).
template <bool MIX, bool NT>
__global__ __launch_bounds__(256, 3) void k_roundeval_fp4(const uint4 *__restrict__ a_hi, const uint4 *__restrict__ a_lo,
                                                          const uint4 *__restrict__ b_hi, const uint4 *__restrict__ b_lo, uint64_t n, f128 *out,
                                                          fin_fuse fz, uint32_t xcd_tiles)
{
	__shared__ __attribute__((aligned(16))) uint32_t T[kTile4W];
	__shared__ uint4 raw[kRawSlots][4][kTP];
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const stage4_role sr = make_stage4_role();
	const gram4_role gr = make_gram4_role(wave, lane);

	v16f acc[kAccTiles];
	acc4_zero(acc);

	const uint64_t n_tiles_all = (n + kTP - 1) / kTP;
	// tile order: XCD x = blockIdx.x & 7 takes the x-th contiguous eighth of the tiles (see kernels_foldeval_mfma.hip)
	uint64_t tbase = 0, tstride = gridDim.x, n_tiles = n_tiles_all, t0 = blockIdx.x;
	if ((xcd_tiles & 1) && (gridDim.x & 7) == 0) {
		const uint64_t chunk = (n_tiles_all + 7) >> 3;
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		n_tiles = tbase >= n_tiles_all ? 0 : (n_tiles_all - tbase < chunk ? n_tiles_all - tbase : chunk);
	}
	// this wave's 64 points of tile t into raw[slot]: four 1-KiB requests (a lane past the end fetches element 0: zeroed below).
	// Inline assembly on purpose: the compiler waits for EVERY outstanding LDS-DMA load before any LDS read or workgroup
	// fence it can see (it cannot tell raw[] from T[]), which would take the three-tile prefetch down to none; here the
	// waits are the explicit s_waitcnt below.
	const uint32_t raw_base = (uint32_t)(uintptr_t)(&raw[0][0][0]) + wave * 64 * 16;
	auto fetch = [&](uint64_t t, unsigned slot) {
		const uint64_t pt = (tbase + t) * kTP + threadIdx.x;
		const uint64_t e = pt < n ? pt : 0;
		const uint32_t l0 = __builtin_amdgcn_readfirstlane(raw_base + slot * (4 * kTP * 16));
		if constexpr (NT) {
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(a_hi + e), "s"(l0) : "memory", "m0");
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(a_lo + e), "s"(l0 + 1 * kTP * 16) : "memory", "m0");
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(b_hi + e), "s"(l0 + 2 * kTP * 16) : "memory", "m0");
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(b_lo + e), "s"(l0 + 3 * kTP * 16) : "memory", "m0");
		} else {
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a_hi + e), "s"(l0) : "memory", "m0");
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a_lo + e), "s"(l0 + 1 * kTP * 16) : "memory", "m0");
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(b_hi + e), "s"(l0 + 2 * kTP * 16) : "memory", "m0");
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(b_lo + e), "s"(l0 + 3 * kTP * 16) : "memory", "m0");
		}
	};
	// workgroup barrier that waits for this wave's LDS traffic only (not for the LDS-DMA loads in flight)
	auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

	// tile t out of the raw slot into registers (each wave reads what its own lanes fetched), zeros past the end; the slot is
	// refilled at once with the tile after it, which then has a whole iteration to arrive
	uint4 x[4];
	auto take = [&](uint64_t tt) {
		if (tt < n_tiles) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
			for (int k = 0; k < 4; k++)
				x[k] = raw[0][k][threadIdx.x];
			if ((tbase + tt + 1) * kTP > n) { // (uniform) the ragged last tile: lanes past the end contribute zeros
				if ((tbase + tt) * kTP + threadIdx.x >= n) {
#pragma unroll
					for (int k = 0; k < 4; k++)
						x[k] = uint4{0, 0, 0, 0};
				}
			}
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			if (tt + tstride < n_tiles) fetch(tt + tstride, 0);
		}
	};
	uint64_t t = t0;
	if (t < n_tiles) {
		fetch(t, 0);
		take(t);
	}
	for (; t < n_tiles; t += tstride) {
		lds_barrier(); // the previous tile's k-steps are done with T
		// the staging phase (VALU) issues ahead of the other workgroups' k-steps (matrix pipe): xcd_tiles bits 1 .. 2 = its priority
		// (BN_FP4_PRIO; round 0 at n = 28 0.500 - 0.504 -> 0.529 - 0.532 of the roofline, experiments/fp4_prio.txt)
		switch ((xcd_tiles >> 1) & 3) {
		case 1: __builtin_amdgcn_s_setprio(1); break;
		case 2: __builtin_amdgcn_s_setprio(2); break;
		case 3: __builtin_amdgcn_s_setprio(3); break;
		default: break;
		}
		stage4_elem(T, sr, 0, x[0]);
		stage4_elem(T, sr, 2, MIX ? uint4{x[0].x ^ x[1].x, x[0].y ^ x[1].y, x[0].z ^ x[1].z, x[0].w ^ x[1].w} : x[1]);
		stage4_elem(T, sr, 1, x[2]);
		stage4_elem(T, sr, 3, MIX ? uint4{x[2].x ^ x[3].x, x[2].y ^ x[3].y, x[2].z ^ x[3].z, x[2].w ^ x[3].w} : x[3]);
		if (xcd_tiles & 6) __builtin_amdgcn_s_setprio(0);
		lds_barrier(); // T staged
		gram4_tile(T, gr, acc);
		take(t + tstride);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	tail4(acc, gr, wave, lane, out, fz, fz.args.seq);
}

// ---- the same evaluation with the two halves of the work on different waves -----------------------------------------------
// (the structure of kernels_foldeval_fp4.hip, without the fold).  A workgroup is 12 waves on one CU: waves 4 .. 11 load the four
// elements of a point straight into registers (two tiles ahead), nibble-transpose them and stage a PAIR of tiles into T[i & 1]
// while waves 0 .. 3 run the k-steps of the previous pair out of T[(i - 1) & 1] -- nine-block roles (gram4k_role: five or four
// accumulator tiles, which leaves room for a second operand register set) with the k-steps software-pipelined, so that a Gram wave
// keeps the matrix pipe of its SIMD busy by itself; one workgroup barrier per pair, no LDS round trip for the raw elements.  Whole
// tiles only, at least two tiles per CU.
namespace {
constexpr unsigned kWsGramWaves = 4, kWsGroups = 2, kWsThreads = 64 * kWsGramWaves * (1 + kWsGroups);
typedef unsigned int ws_v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ws_load(const uint4 *p)
{
	if constexpr (NT) {
		const ws_v4u v = __builtin_nontemporal_load(reinterpret_cast<const ws_v4u *>(p));
		return uint4{v.x, v.y, v.z, v.w};
	} else {
		return *p;
	}
}
} // namespace

template <bool MIX, bool NT>
__global__ __launch_bounds__(kWsThreads, 1) void k_roundeval_fp4_ws(const uint4 *__restrict__ a_hi, const uint4 *__restrict__ a_lo,
                                                                    const uint4 *__restrict__ b_hi, const uint4 *__restrict__ b_lo, uint64_t n,
                                                                    f128 *out, fin_fuse fz, uint32_t xcd_tiles)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t T_dyn[]; // 2 buffers x 2 tiles of FP4 operands
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool stages = wave >= kWsGramWaves;
	const unsigned grp = stages ? (wave - kWsGramWaves) >> 2 : 0;  // which tile of the pair
	const unsigned stid = (threadIdx.x - 64 * kWsGramWaves) & 255; // the lane's point inside its tile
	const uint64_t n_tiles_all = n / kTP;
	uint64_t tbase = 0, tstride = gridDim.x, tlimit = n_tiles_all, t0 = blockIdx.x;
	if ((xcd_tiles & 1) && (gridDim.x & 7) == 0) {
		const uint64_t chunk = (n_tiles_all + 7) >> 3;
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		tlimit = tbase >= n_tiles_all ? 0 : (n_tiles_all - tbase < chunk ? n_tiles_all - tbase : chunk);
	}
	// pairs of this workgroup: pair p = tiles t0 + (2 p + g) * tstride, g = 0, 1
	const uint64_t n_seq = t0 < tlimit ? (tlimit - t0 + tstride - 1) / tstride : 0;
	const uint64_t n_pairs = (n_seq + 1) >> 1;
	if (stages) {
		__builtin_amdgcn_s_setprio(3); // the stager waves issue ahead of their SIMD's Gram wave
		const stage4_role sr = make_stage4_role(stid);
		uint4 xa[4], xb[4]; // the group's tile of pair p (even p: xa, odd p: xb), loaded two pairs ahead
		auto load = [&](uint4 (&x)[4], uint64_t p) {
			const uint64_t seq = 2 * p + grp;
			if (seq < n_seq) { // (uniform)
				const uint64_t e = (tbase + t0 + seq * tstride) * kTP + stid;
				x[0] = ws_load<NT>(a_hi + e);
				x[1] = ws_load<NT>(a_lo + e);
				x[2] = ws_load<NT>(b_hi + e);
				x[3] = ws_load<NT>(b_lo + e);
			}
		};
		auto half = [&](uint4 (&x)[4], uint64_t p) {
			uint32_t *const Tn = T_dyn + ((p & 1) * kWsGroups + grp) * kTile4kW;
			if (2 * p + grp < n_seq) { // (uniform; false only for group 1 on an odd last pair)
				stage4k_elem(Tn, sr, 0, x[0]);
				stage4k_elem(Tn, sr, 2, MIX ? uint4{x[0].x ^ x[1].x, x[0].y ^ x[1].y, x[0].z ^ x[1].z, x[0].w ^ x[1].w} : x[1]);
				stage4k_elem(Tn, sr, 1, x[2]);
				stage4k_elem(Tn, sr, 3, MIX ? uint4{x[2].x ^ x[3].x, x[2].y ^ x[3].y, x[2].z ^ x[3].z, x[2].w ^ x[3].w} : x[3]);
			}
			load(x, p + 2);
			__syncthreads(); // the pair is staged; the Gram waves are done with the buffer this wave writes next
		};
		load(xa, 0);
		load(xb, 1);
		for (uint64_t p = 0; p < n_pairs; p += 2) {
			half(xa, p);
			if (p + 1 < n_pairs) half(xb, p + 1);
		}
	} else {
		const gram4k_role gr = make_gram4k_role(wave >> 1, (wave ^ blockIdx.x) & 1, lane);
		auto run = [&](auto qc) {
			constexpr int Q = decltype(qc)::value;
			v16f acc[Q == 0 ? 5 : 4];
#pragma unroll
			for (int i = 0; i < (Q == 0 ? 5 : 4); i++)
#pragma unroll
				for (int r = 0; r < 16; r++)
					acc[i][r] = 0.0f;
			for (uint64_t p = 0; p < n_pairs; p++) {
				__syncthreads();
				const uint32_t *Tp = T_dyn + (p & 1) * kWsGroups * kTile4kW;
				const int n_t = 2 * p + 1 < n_seq ? 2 : 1; // (uniform) both tiles of the pair, or the odd last one
				for (int i = 0; i < n_t; i++)
					gram4k_steps<Q, 4>(Tp + i * kTile4kW, gr, acc);
			}
			__syncthreads(); // every wave is done with the operand tiles: the parity words take their place
			parity4k<Q>(acc, gr, lane, *reinterpret_cast<gram4k_parity *>(T_dyn));
		};
		if (gr.q == 0) // (uniform per wave)
			run(std::integral_constant<int, 0>{});
		else
			run(std::integral_constant<int, 1>{});
	}
	if (stages) __syncthreads(); // (the Gram waves' barrier in front of the parity words)
	__shared__ uint64_t z3[2][3];
	sums4k(*reinterpret_cast<gram4k_parity *>(T_dyn), wave, lane, z3);
	tail_publish(z3, out, fz, fz.args.seq, nullptr);
}

template <bool MIX>
static hipError_t launch_fp4(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo, uint64_t n, f128 *d_out,
                             const fin_fuse *fuse)
{
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	if (n == 0) return hipErrorNotSupported;
	const uint64_t n_tiles = (n + kTP - 1) / kTP;
	const uint64_t cap = (uint64_t)n_cu * 3; // three workgroups per CU (47 KiB of LDS, <= 168 registers)
	const unsigned grid = (unsigned)(n_tiles < cap ? n_tiles : cap);
	if ((n_tiles + grid - 1) / grid > (1ull << 14)) return hipErrorNotSupported; // 2^22 points per workgroup: the f32 counts stay exact
	static const uint32_t xcd_tiles = [] {
		const char *e = bn::settled_knob("BN_XCD_TILES");
		const char *p = bn::settled_knob("BN_FP4_PRIO");
		return (uint32_t)!(e && e[0] == '0') | (((p ? (uint32_t)atoi(p) : 2u) & 3u) << 1); // BN_FP4_PRIO=0 .. 3 (default 2)
	}();
	// BN_FP4_NT_MIN_LOG2: points from which the loads are non-temporal (measurement knob; 64 = never)
	static const int nt_min_log2 = [] {
		const char *e = bn::settled_knob("BN_FP4_NT_MIN_LOG2");
		return e ? atoi(e) : 24;
	}();
	const bool nt = nt_min_log2 < 64 && n >= (1ull << nt_min_log2);
	// BN_FP4_WS=0: never the wave-specialised form; BN_FP4_WS_MIN_LOG2: points from which it takes the launch
	static const int ws_min_log2 = [] {
		const char *e = getenv("BN_FP4_WS");
		if (e && e[0] == '0') return 64;
		const char *m = bn::settled_knob("BN_FP4_WS_MIN_LOG2");
		return m ? atoi(m) : 20;
	}();
	if (ws_min_log2 < 64 && n >= (1ull << ws_min_log2) && n % kTP == 0 && n_tiles >= 2 * (uint64_t)n_cu && (n_tiles + n_cu - 1) / n_cu <= (1ull << 14)) {
		constexpr unsigned lds = 2 * kWsGroups * kTile4kW * 4;
		// BN_FP4_WS_GRID=g (tests): g workgroups instead of one per CU -- odd tile counts per workgroup and the plain tile order, which
		// 256 CUs and power-of-two sizes never produce
		static const unsigned grid_override = [] {
			const char *e = getenv("BN_FP4_WS_GRID");
			return e ? (unsigned)atoi(e) : 0u;
		}();
		unsigned ws_grid = (unsigned)n_cu;
		if (grid_override && grid_override <= ws_grid && (n_tiles + grid_override - 1) / grid_override <= (1ull << 14)) ws_grid = grid_override;
		const hipError_t attr = [] {
			hipError_t e = func_lds_limit(reinterpret_cast<const void *>(&k_roundeval_fp4_ws<MIX, true>), lds);
			if (e != hipSuccess) return e;
			return func_lds_limit(reinterpret_cast<const void *>(&k_roundeval_fp4_ws<MIX, false>), lds);
		}();
		if (attr != hipSuccess) return attr;
		if (nt)
			hipLaunchKernelGGL((k_roundeval_fp4_ws<MIX, true>), dim3(ws_grid), dim3(kWsThreads), lds, s, (const uint4 *)a_hi, (const uint4 *)a_lo,
			                   (const uint4 *)b_hi, (const uint4 *)b_lo, n, d_out, fz, xcd_tiles & 1u);
		else
			hipLaunchKernelGGL((k_roundeval_fp4_ws<MIX, false>), dim3(ws_grid), dim3(kWsThreads), lds, s, (const uint4 *)a_hi, (const uint4 *)a_lo,
			                   (const uint4 *)b_hi, (const uint4 *)b_lo, n, d_out, fz, xcd_tiles & 1u);
		return hipGetLastError();
	}
	if (nt)
		hipLaunchKernelGGL((k_roundeval_fp4<MIX, true>), dim3(grid), dim3(256), 0, s, (const uint4 *)a_hi, (const uint4 *)a_lo, (const uint4 *)b_hi,
		                   (const uint4 *)b_lo, n, d_out, fz, xcd_tiles);
	else
		hipLaunchKernelGGL((k_roundeval_fp4<MIX, false>), dim3(grid), dim3(256), 0, s, (const uint4 *)a_hi, (const uint4 *)a_lo, (const uint4 *)b_hi,
		                   (const uint4 *)b_lo, n, d_out, fz, xcd_tiles);
	return hipGetLastError();
}

// d_out[0] ^= sum_i a_hi[i]*b_hi[i] ; d_out[1] ^= sum_i (a_lo[i]^a_hi[i])*(b_lo[i]^b_hi[i])
hipError_t launch_roundeval_fp4_pair(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo, uint64_t n,
                                     f128 *d_out, const fin_fuse *fuse)
{
	return launch_fp4<true>(s, n_cu, a_hi, a_lo, b_hi, b_lo, n, d_out, fuse);
}

// d_out[0] ^= sum_{i<n} a[i]*b[i] ; d_out[1] ^= sum_{i<n} a[i+split]*b[i+split]
hipError_t launch_roundeval_fp4_split(hipStream_t s, int n_cu, const void *a, const void *b, uint64_t n, uint64_t split_off, f128 *d_out)
{
	const char *a2 = (const char *)a + split_off * 16, *b2 = (const char *)b + split_off * 16;
	return launch_fp4<false>(s, n_cu, a, a2, b, b2, n, d_out, nullptr);
}

} // namespace bn
