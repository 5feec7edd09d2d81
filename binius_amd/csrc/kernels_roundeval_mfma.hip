// binius_amd/csrc/kernels_roundeval_mfma.hip -- round evaluation of the bivariate product on the matrix
// cores (gram.hpp):
//   S_1 = sum_i a_hi[i]*b_hi[i],   S_inf = sum_i (a_lo[i]+a_hi[i])*(b_lo[i]+b_hi[i])
// (crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-408), one pass over the data; the
// SPLIT form computes two plain sums of products (inner_product / sum-of-products compositions).
//
// Workgroup = 4 waves, two workgroups per CU (two waves per SIMD; the workgroups drift apart, so one
// stages while the other multiplies).  Per tile of 256 points a lane loads the four elements of ONE point
// (a_hi, a_lo, b_hi, b_lo: 16-byte coalesced loads, the next tile's elements fly while this tile is
// multiplied), byte-transposes them across its quad into the LDS tile (stage_T), and after the barrier
// its wave runs the Gram k-steps of its (product, column half).
// Algorithmic bytes: 16*m*n per launch (m = 2 arrays, n points), read once.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "gram.hpp"

namespace bn {

using namespace gram;

// SA, SB: log2 of the element stride of the a / b operands (1: the pairs of the old HAL's Low-to-High order, interleaved)
template <bool SPLIT, int SA = 0, int SB = 0>
__global__ __launch_bounds__(256, 2) void k_roundeval_mfma(const uint4 *__restrict__ a_hi, const uint4 *__restrict__ a_lo,
                                                           const uint4 *__restrict__ b_hi, const uint4 *__restrict__ b_lo, uint64_t n,
                                                           f128 *out, fin_fuse fz, uint32_t xcd_tiles)
{
	constexpr int kTiles = 2; // tiles per iteration: 32 KiB of loads in flight per workgroup, one barrier per 512 points
	__shared__ __attribute__((aligned(16))) uint32_t T[2][kTiles][kTileW];
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const stage_role sr = make_stage_role(wave, lane);
	const gram_role gr = make_gram_role(wave, lane);

	v16i acc[kAccTiles];
	acc_zero(acc);

	const uint64_t n_groups_all = (n + kTiles * kTP - 1) / (kTiles * kTP);
	// group order: XCD x = blockIdx.x & 7 takes the x-th contiguous eighth of the groups (see kernels_foldeval_mfma.hip)
	uint64_t gbase = 0, gstride = gridDim.x, n_groups = n_groups_all, g0 = blockIdx.x;
	if (xcd_tiles && (gridDim.x & 7) == 0) {
		const uint64_t chunk = (n_groups_all + 7) >> 3;
		gbase = (blockIdx.x & 7) * chunk;
		gstride = gridDim.x >> 3;
		g0 = blockIdx.x >> 3;
		n_groups = gbase >= n_groups_all ? 0 : (n_groups_all - gbase < chunk ? n_groups_all - gbase : chunk);
	}
	// a_hi, a_lo, b_hi, b_lo of this lane's point in each tile of the group.  One tile of 16 KiB in flight
	// per workgroup covers only ~2.7 us of HBM latency at 3 TB/s -- what a loaded HBM takes to answer.
	// (a lane past the end loads element 0 and zeroes it when the tile is staged: nothing depends on a
	// loaded value before then, so the loads stay in flight across the Gram k-steps)
	uint4 x[kTiles][4];
	auto load = [&](uint64_t g) {
#pragma unroll
		for (int i = 0; i < kTiles; i++) {
			const uint64_t pt = ((gbase + g) * kTiles + i) * kTP + threadIdx.x;
			const uint64_t e = pt < n ? pt : 0;
			x[i][0] = a_hi[e << SA];
			x[i][1] = a_lo[e << SA];
			x[i][2] = b_hi[e << SB];
			x[i][3] = b_lo[e << SB];
		}
	};
	auto stage = [&](uint64_t g, uint32_t (*Tb)[kTileW]) {
#pragma unroll
		for (int i = 0; i < kTiles; i++) {
			if (((gbase + g) * kTiles + i) * kTP + threadIdx.x >= n) {
#pragma unroll
				for (int k = 0; k < 4; k++)
					x[i][k] = uint4{0, 0, 0, 0};
			}
			stage_T<!SPLIT>(Tb[i], sr, 0, x[i][0], x[i][1]);
			stage_T<!SPLIT>(Tb[i], sr, 1, x[i][2], x[i][3]);
		}
	};

	uint64_t g = g0;
	unsigned buf = 0;
	if (g < n_groups) {
		load(g);
		stage(g, T[0]);
	}
	__syncthreads();
	for (; g < n_groups; g += gstride) {
		const uint64_t gn = g + gstride;
#ifndef GRAM_DBG
#define GRAM_DBG 0
#endif
		if (!(GRAM_DBG & 1) && gn < n_groups) load(gn);
#pragma unroll
		for (int i = 0; i < kTiles; i++)
			gram_tile(T[buf][i], gr, acc);
		if (!(GRAM_DBG & 2) && gn < n_groups) stage(gn, T[buf ^ 1]);
		if (!(GRAM_DBG & 4)) __syncthreads();
		buf ^= 1;
	}
	gram::tail(acc, wave, lane, out, fz, fz.args.seq);
}

bool mfma_applies(int n_cu, uint64_t n_points)
{
	// BN_EVAL=valu: 9-lane VALU kernels only; BN_MFMA_MIN_TILES: tiles from which the matrix-core kernels run
	static const int64_t min_tiles = [] {
		const char *m = getenv("BN_EVAL");
		if (m && m[0] == 'v') return (int64_t)-1;
		const char *e = bn::settled_knob("BN_MFMA_MIN_TILES");
		return e ? (int64_t)atoll(e) : (int64_t)0;
	}();
	if (min_tiles < 0) return false;
	const uint64_t n_tiles = (n_points + kTP - 1) / kTP;
	// one tile per CU is enough (round 4, tools/r04_min_tiles.sh: r = 18 of a sumcheck on the matrix cores instead of the 9-lane
	// kernel: n = 24 0.755 - 0.79 -> 0.734 - 0.749 ms, n = 20 0.228 -> 0.2255; half a tile per CU gains nothing more)
	return n_tiles >= (min_tiles ? (uint64_t)min_tiles : (uint64_t)n_cu);
}

static unsigned grid_mfma(uint64_t n, int n_cu)
{
	const uint64_t n_groups = (n + 2 * kTP - 1) / (2 * kTP); // kTiles = 2 tiles per iteration
	const uint64_t cap = (uint64_t)n_cu * 2;
	return (unsigned)(n_groups < cap ? (n_groups ? n_groups : 1) : cap);
}

template <bool SPLIT>
static hipError_t launch_mfma(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo, uint64_t n,
                              f128 *d_out, const fin_fuse *fuse)
{
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	if (n == 0 && fuse) return hipErrorNotSupported; // nothing to launch: the caller finalizes separately
	if (n == 0) return hipSuccess;
	static const uint32_t xcd_tiles = [] {
		const char *e = bn::settled_knob("BN_XCD_TILES");
		return (uint32_t)!(e && e[0] == '0');
	}();
	hipLaunchKernelGGL((k_roundeval_mfma<SPLIT>), dim3(grid_mfma(n, n_cu)), dim3(256), 0, s, (const uint4 *)a_hi, (const uint4 *)a_lo,
	                   (const uint4 *)b_hi, (const uint4 *)b_lo, n, d_out, fz, xcd_tiles);
	return hipGetLastError();
}

// Evaluations of 2^20 points and more go to the FP4 matrix path (kernels_roundeval_fp4.hip): twice the k-depth per
// instruction, half the operand masks per point, element loads through LDS-DMA, three workgroups per CU.  BN_FP4=0: the int8
// kernel at every size; BN_FP4_MIN_LOG2: smallest log2(points) for the FP4 kernel.
static bool fp4_applies(uint64_t n)
{
	static const int fp4_min_log2 = [] {
		const char *e = getenv("BN_FP4");
		if (e && e[0] == '0') return 64;
		const char *m = getenv("BN_FP4_MIN_LOG2");
		return m ? atoi(m) : 20; // measured: equal at 2^19 points, 10 - 25 % faster from 2^20 on (profiles/r02/fp4_crossover.txt)
	}();
	return fp4_min_log2 < 64 && n >= ((uint64_t)1 << fp4_min_log2);
}

// d_out[0] ^= sum_i a_hi[i]*b_hi[i] ; d_out[1] ^= sum_i (a_lo[i]^a_hi[i])*(b_lo[i]^b_hi[i])
hipError_t launch_roundeval_mfma_pair(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo,
                                      uint64_t n, f128 *d_out, const fin_fuse *fuse)
{
	if (fp4_applies(n)) {
		const hipError_t e = launch_roundeval_fp4_pair(s, n_cu, a_hi, a_lo, b_hi, b_lo, n, d_out, fuse);
		if (e != hipErrorNotSupported) return e;
	}
	return launch_mfma<false>(s, n_cu, a_hi, a_lo, b_hi, b_lo, n, d_out, fuse);
}

hipError_t launch_roundeval_mfma_pair_strided(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, uint32_t a_shift, const void *b_hi,
                                              const void *b_lo, uint32_t b_shift, uint64_t n, f128 *d_out)
{
	if (n == 0) return hipSuccess;
	if (a_shift > 1 || b_shift > 1) return hipErrorNotSupported;
	const fin_fuse fz{};
	const dim3 grid(grid_mfma(n, n_cu)), block(256);
#define BN_LAUNCH_STRIDED(SA_, SB_)                                                                                                              \
	hipLaunchKernelGGL((k_roundeval_mfma<false, SA_, SB_>), grid, block, 0, s, (const uint4 *)a_hi, (const uint4 *)a_lo, (const uint4 *)b_hi, \
	                   (const uint4 *)b_lo, n, d_out, fz, 1u)
	if (a_shift == 1 && b_shift == 1) BN_LAUNCH_STRIDED(1, 1);
	else if (a_shift == 1) BN_LAUNCH_STRIDED(1, 0);
	else if (b_shift == 1) BN_LAUNCH_STRIDED(0, 1);
	else BN_LAUNCH_STRIDED(0, 0);
#undef BN_LAUNCH_STRIDED
	return hipGetLastError();
}

// d_out[0] ^= sum_{i<n} a[i]*b[i] ; d_out[1] ^= sum_{i<n} a[i+split]*b[i+split]
hipError_t launch_roundeval_mfma_split(hipStream_t s, int n_cu, const void *a, const void *b, uint64_t n, uint64_t split_off, f128 *d_out)
{
	if (fp4_applies(n)) {
		const hipError_t e = launch_roundeval_fp4_split(s, n_cu, a, b, n, split_off, d_out);
		if (e != hipErrorNotSupported) return e;
	}
	const char *a2 = (const char *)a + split_off * 16, *b2 = (const char *)b + split_off * 16;
	return launch_mfma<true>(s, n_cu, a, a2, b, b2, n, d_out, nullptr);
}

} // namespace bn
