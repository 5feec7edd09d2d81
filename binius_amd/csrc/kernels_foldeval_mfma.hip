// binius_amd/csrc/kernels_foldeval_mfma.hip -- fold of round r and round evaluation of round r+1 in ONE
// pass over the data, the evaluation on the matrix cores (gram.hpp):
//
//   a'[i] = a[i] + z*(a[i + N/2] - a[i])        i < N/2        (extrapolate_line, layer.rs:421)
//   S_1   = sum_{j < N/4} a'[j + N/4] * b'[j + N/4]
//   S_inf = sum_{j < N/4} (a'[j] + a'[j + N/4]) * (b'[j] + b'[j + N/4])
//                                                (v3/bivariate_product.rs:217-228 then :303-408)
//
// Same contract as kernels_foldeval9.hip (which keeps the small rounds): the ABI defers the fold and,
// when the next kernel launch evaluates exactly the folded arrays, runs this kernel instead of two.
// Algorithmic bytes: read 16*m*N + write 8*m*N = 24*m*N per launch.
//
// Workgroup = 4 waves, two workgroups per CU.  Per tile of 256 evaluation points a lane folds the four
// elements of ONE point (array x half): 16-byte coalesced loads (prefetched one tile ahead), the
// constant multiplication through the LDS nibble tables (ctable.hpp), a 16-byte coalesced store of the
// folded element (it is the next round's input), and the quad byte transposes that turn the folded
// registers into the point's Gram-tile words (stage_T) -- the folded values never take a second trip
// through memory.  The Gram k-steps of the PREVIOUS tile are issued between the four constant
// multiplications: matrix pipe and VALU/LDS work side by side; one workgroup barrier per tile.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "arm.hpp"
#include "ctable.hpp"
#include "gram.hpp"

namespace bn {

using namespace gram;

// FE_ABL (measurement builds of tools/gram_bench.hip only; the library is built with 0): bit 0 = no Gram k-steps, bit 1 = the fold
// without its constant multiplication, bit 2 / 3 = non-temporal loads / stores, bit 4 = no staging of the folded registers,
// bit 5 = no workgroup barrier per tile (only with bits 0 and 4).  What the memory system gives the kernel's access pattern
// with the arithmetic taken out piece by piece (tools/r04_fe_ablation.sh).
#ifndef FE_ABL
#define FE_ABL 0
#endif
typedef unsigned int fe_v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 fe_load(const uint4 *p)
{
	if constexpr (NT || (FE_ABL & 4) != 0) {
		const fe_v4u v = __builtin_nontemporal_load(reinterpret_cast<const fe_v4u *>(p));
		return uint4{v.x, v.y, v.z, v.w};
	} else {
		return *p;
	}
}
template <bool NT>
__device__ __forceinline__ void fe_store(uint4 *p, uint4 r)
{
	if constexpr (NT || (FE_ABL & 8) != 0) {
		const fe_v4u v = {r.x, r.y, r.z, r.w};
		__builtin_nontemporal_store(v, reinterpret_cast<fe_v4u *>(p));
	} else {
		*p = r;
	}
}

// SC: 0 = plain fold; 1 / 2 = the upper half of folded array 0 / 1 is multiplied by fa.hi_scale (a fifth constant
// multiplication per point, through a second nibble table) before it is stored and staged.
// FULL: the number of evaluation points is a multiple of the tile (every size a prover reaches this kernel with): no lane is ever
// past the end, so the stores are unconditional.  This is not cosmetic: behind a conditional block of stores the compiler's wait
// for a quadrant's loads has to hold on the path that skipped the stores as well -- `s_waitcnt vmcnt(6)` where the path with the
// stores allows vmcnt(10) --, and a wave then waits for the loads of the next TWO quadrants too, issued half an iteration ago
// instead of a whole one.
// NT: non-temporal loads and stores -- the arrays of an HBM-resident round are touched once per launch, and marking the accesses
// as streaming is worth 2 % there (profiles/r04/fe_full_nt.txt); the cache-resident rounds keep the plain accesses (the folded
// halves ARE the next round's input).
template <int SC, bool FULL, bool NT>
__global__ __launch_bounds__(256, 2) void k_foldeval_mfma(foldeval_args fa, uint64_t n_in, f128 z, f128 *out, fin_fuse fz, arm_args arm)
{
	__shared__ __attribute__((aligned(16))) uint32_t T[2][kTileW];
	__shared__ ctable_smem tab;
	__shared__ ctable_opt<SC != 0> tab_hs;
	__shared__ fin_cache fcache;
	const uint64_t seq = fz.args.seq;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const stage_role sr = make_stage_role(wave, lane);
	const gram_role gr = make_gram_role(wave, lane);
	const uint64_t n = n_in >> 2; // evaluation points of the next round
	const uint64_t n_tiles = (n + kTP - 1) / kTP;

	v16i acc[kAccTiles];
	acc_zero(acc);
	BN_TS(0);

	// quadrant k = 2 * array + half: element index half * n + point
	uint4 x0[4], x1[4];
	// (a lane past the end loads element 0 of the quadrant: its fold result is zeroed before use, so
	// nothing depends on the loaded value until the constant multiplication consumes it)
	// Tile order.  Consecutive workgroup ids go to different XCDs (round robin), so XCD x = b & 7 takes the x-th
	// contiguous eighth of the tiles and its 64 workgroups stride through that eighth: at any time an XCD (its L2, its
	// share of the fabric) works on twelve contiguous 256 KiB windows instead of every eighth tile of twelve 2 MiB
	// windows.  +1 ... +2.7 % on the large launches at n = 28, depending on the box (A/B runs alternating on three).
	// BN_XCD_TILES=0: workgroup b takes tiles b, b + G, ... .
	uint64_t tbase = 0, tstride = gridDim.x, tlimit = n_tiles, t0 = blockIdx.x;
	if (fa.xcd_tiles && (gridDim.x & 7) == 0) {
		const uint64_t chunk = (n_tiles + 7) >> 3;
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		tlimit = tbase >= n_tiles ? 0 : (n_tiles - tbase < chunk ? n_tiles - tbase : chunk);
	}
	auto load1 = [&](uint64_t t, int k) {
		const uint64_t pt = (tbase + t) * kTP + threadIdx.x;
		const uint64_t e = (k & 1 ? n : 0) + (FULL || pt < n ? pt : 0);
		x0[k] = fe_load<NT>((const uint4 *)fa.x0[k >> 1] + e);
		x1[k] = fe_load<NT>((const uint4 *)fa.x1[k >> 1] + e);
	};
	auto load = [&](uint64_t t) {
#pragma unroll
		for (int k = 0; k < 4; k++)
			load1(t, k);
	};
	uint64_t t = t0;
	if (t < tlimit) load(t);
	BN_TS(1);
	{
		// the finalize arguments travel with the first tile and wait in LDS for the tail (finalize.hpp)
		const fin_pref fpre = fin_prefetch(fz);
		if (arm.h_cmd) { // (uniform) armed launch of a mid-size round: the challenge arrives through the command block (arm.hpp)
			f128 hs_in;
			if (!arm_wait(arm, z, hs_in)) return;
			fa.hi_scale = hs_in;
		}
		if constexpr (SC != 0) { // both tables side by side (two halves of the workgroup): a build is one dependent chain
			const unsigned grp = threadIdx.x >> 7;
			ctable_build_group(grp ? tab_hs.get() : tab, grp ? fa.hi_scale : z, threadIdx.x & 127, 128);
		} else {
			ctable_build(tab, z); // the loads above are in flight meanwhile; ends with a barrier
		}
		fin_commit(fz, fpre, fcache);
	}
	BN_TS(2);

	// One iteration: fold tile tt (VALU + LDS lookups) with, when GRAM, the Gram k-steps of the previous
	// tile (in Tp) between the four constant multiplications (matrix pipe); the folded registers go to HBM
	// and, byte-transposed, into the Gram tile Tn.
	auto iteration = [&](uint64_t tt, const uint32_t *Tp, uint32_t *Tn, auto with_gram) {
		constexpr bool GRAM = decltype(with_gram)::value && !(FE_ABL & 1);
		const uint64_t pt = (tbase + tt) * kTP + threadIdx.x;
		const bool ok = FULL || pt < n;
		// the last iteration re-requests its own tile (cache hits) instead of branching around the loads
		const uint64_t tn = tt + tstride < tlimit ? tt + tstride : tt;
		gram_pipe gp;
		uint4 f[4];
		if (GRAM) gram_begin(Tp, gr, gp);
		// FE_VARIANT (measurement knob, tools/r04_fe_variants.sh): 0 = round 3's form (x0 ^ ctable_mul_pinned<4>); bit 0 = the
		// "x0 +" of the fold rides in the product's first three-input XOR, bit 1 = table offsets formed per group of lookups
		// (4 live offset registers instead of 32), bit 2 = groups of 8 lookups in flight.  7: 813 instead of 841 VALU per tile
		// and wave, 252 registers, no scratch; 0.7523 -> 0.7415 ms at 2^26, 1.4832 -> 1.4653 ms at 2^27 elements per array
		// (sustained launches, two alternating runs, profiles/r04/fe_variants.txt).
#ifndef FE_VARIANT
#define FE_VARIANT 7
#endif
#if FE_ABL & 2
#define FE_FOLD(k) xor4(x0[k], x1[k])
#elif FE_VARIANT == 0
#define FE_FOLD(k) xor4(x0[k], ctable_mul_pinned<4>(tab, xor4(x0[k], x1[k])))
#else
#define FE_FOLD(k) ctable_mul_acc<(FE_VARIANT & 4) ? 8 : 4, (FE_VARIANT & 2) != 0>(tab, xor4(x0[k], x1[k]), x0[k])
#endif
		f[0] = FE_FOLD(0);
		load1(tn, 0); // this quadrant of the next tile flies from here on
		if (GRAM) {
			gram_step<0>(Tp, gr, gp, acc);
			gram_step<1>(Tp, gr, gp, acc);
		}
		f[1] = FE_FOLD(1);
		load1(tn, 1); // this quadrant of the next tile flies from here on
		if constexpr (SC == 1) f[1] = ctable_mul_pinned<4>(tab_hs.get(), f[1]);
		if (GRAM) {
			gram_step<2>(Tp, gr, gp, acc);
			gram_step<3>(Tp, gr, gp, acc);
		}
		f[2] = FE_FOLD(2);
		load1(tn, 2); // this quadrant of the next tile flies from here on
		if (GRAM) {
			gram_step<4>(Tp, gr, gp, acc);
			gram_step<5>(Tp, gr, gp, acc);
		}
		f[3] = FE_FOLD(3);
		load1(tn, 3); // this quadrant of the next tile flies from here on
		if constexpr (SC == 2) f[3] = ctable_mul_pinned<4>(tab_hs.get(), f[3]);
		if (GRAM) {
			gram_step<6>(Tp, gr, gp, acc);
			gram_step<7>(Tp, gr, gp, acc);
		}
		if (ok) {
#pragma unroll
			for (int k = 0; k < 4; k++)
				fe_store<NT>((uint4 *)fa.out[k >> 1] + ((k & 1 ? n : 0) + pt), f[k]);
		} else {
#pragma unroll
			for (int k = 0; k < 4; k++)
				f[k] = uint4{0, 0, 0, 0};
		}
		// quadrant 2 * array + half: half 1 is the evaluation at 1, half 0 its partner; points past the
		// end carry zeros
		if constexpr (!(FE_ABL & 16)) {
			stage_T<true>(Tn, sr, 0, f[1], f[0]);
			stage_T<true>(Tn, sr, 1, f[3], f[2]);
		}
		if constexpr (!(FE_ABL & 32)) __syncthreads();
	};
	if (t < tlimit) {
		unsigned buf = 0;
		iteration(t, T[1], T[0], std::false_type{});
		BN_TS(3);
		for (t += tstride; t < tlimit; t += tstride) {
			iteration(t, T[buf], T[buf ^ 1], std::true_type{});
			buf ^= 1;
		}
		BN_TS(4);
		if constexpr (!(FE_ABL & 1)) gram_tile(T[buf], gr, acc);
	}
	BN_TS(5);
	gram::tail(acc, wave, lane, out, fz, seq, &fcache);
	BN_TS(12);
}

// For both arrays j: out_j[i] = x0_j[i] + z * (x1_j[i] - x0_j[i]), i < n_in/2 (out_j may be x0_j), and
// accumulate the next round's (S_1, S_inf) of out_0 * out_1 into d_out[0], d_out[1].  n_in >= 4.
hipError_t launch_foldeval_mfma(hipStream_t s, int n_cu, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse *fuse,
                                const arm_args *armed)
{
	arm_args arm{};
	if (armed) arm = *armed;
	if (n_in < 4 || (n_in & 3)) return hipErrorNotSupported;
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	const uint64_t n_tiles = ((n_in >> 2) + kTP - 1) / kTP;
	const uint64_t cap = (uint64_t)n_cu * 2;
	const dim3 grid((unsigned)(n_tiles < cap ? n_tiles : cap));
	static const uint32_t xcd_tiles = [] {
		const char *e = bn::settled_knob("BN_XCD_TILES");
		return (uint32_t)!(e && e[0] == '0');
	}();
	foldeval_args fx = fa;
	fx.xcd_tiles = xcd_tiles;
	const bool full = ((n_in >> 2) % kTP) == 0;
	// BN_FE_NT_MIN_LOG2: elements per array from which the accesses are non-temporal (measurement knob; 64 = never)
	static const int nt_min_log2 = [] {
		const char *e = bn::settled_knob("BN_FE_NT_MIN_LOG2");
		return e ? atoi(e) : 25;
	}();
	const bool nt = full && nt_min_log2 < 64 && n_in >= (1ull << nt_min_log2);
	// whole tiles, plain fold, at least a tile per CU: the wave-specialised FP4 form (kernels_foldeval_fp4.hip)
	if (foldeval_fp4_applies(n_cu, fx, n_in)) return launch_foldeval_fp4(s, n_cu, fx, n_in, z, d_out, fz, arm, nt);
#define BN_FE_LAUNCH(SC)                                                                                                       \
	if (nt)                                                                                                                    \
		hipLaunchKernelGGL((k_foldeval_mfma<SC, true, true>), grid, dim3(256), 0, s, fx, n_in, z, d_out, fz, arm);             \
	else if (full)                                                                                                             \
		hipLaunchKernelGGL((k_foldeval_mfma<SC, true, false>), grid, dim3(256), 0, s, fx, n_in, z, d_out, fz, arm);            \
	else                                                                                                                       \
		hipLaunchKernelGGL((k_foldeval_mfma<SC, false, false>), grid, dim3(256), 0, s, fx, n_in, z, d_out, fz, arm);
	switch (fa.scale_mask) {
	case 0: BN_FE_LAUNCH(0) break;
	case 1: BN_FE_LAUNCH(1) break;
	case 2: BN_FE_LAUNCH(2) break;
	default: return hipErrorNotSupported; // both arrays scaled: the caller runs fold, scale and evaluation separately
	}
#undef BN_FE_LAUNCH
	return hipGetLastError();
}

} // namespace bn
