// binius_amd/csrc/kernels_roundeval9.hip -- the hot round-evaluation kernel for bivariate products:
//   S_1 = sum_i a_hi[i]*b_hi[i],   S_inf = sum_i (a_lo[i]+a_hi[i])*(b_lo[i]+b_hi[i])
// (crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-408), one pass over the data.
//
// Mapping to a wave64 ("9-lane Karatsuba groups").  Two Karatsuba levels split a GF(2^128) product
// into 9 GF(2^32) products of limb combinations
//     {0} {1} {0,1} {2} {3} {2,3} {0,2} {1,3} {0,1,2,3}            (limb = 32-bit word of an element)
// and every post-processing step of Karatsuba is linear, so it commutes with the sum over i: each
// lane owns ONE combination, multiplies it bit-sliced (bs_mul<5>, 32 planes, ~1200 VALU) and keeps a
// private 32-plane accumulator for the whole kernel; the recombination into a field element happens
// once per workgroup.  A wave holds 7 such groups (63 lanes); a batch is 16 hypercube points per
// group.  Registers: operands 2x32 + product 32 + accumulator 32 + temporaries -> two waves per SIMD
// with no scratch, instead of one spilling wave for a monolithic 128-plane product.
//
// Per batch and group, 8 lanes each load one 32-bit word column (a or b, word 0..3) of the 16 points
// -- 16 rows from the hi half and 16 rows from the lo half -- and transpose it in registers
// (32x32 bit transpose); "lo + hi" is then one shifted XOR per plane, which puts the evaluation-at-1
// operands in the low 16 bits and the evaluation-at-infinity operands in the high 16 bits of every
// plane.  The 8 transposed limbs are exchanged through a per-wave LDS tile (no workgroup barrier:
// a wave's DS operations execute in order), each lane XORs together the limbs of its combination.
#include <hip/hip_runtime.h>

#include "re9.hpp"

namespace bn {

using namespace re9;

// SPLIT == false: rows 16..31 come from the lo arrays and planes become [hi | lo^hi]
// SPLIT == true : rows 16..31 come from the same arrays at +split_off and planes stay [x | y]
//                 (a plain sum of products over 2n elements, both halves of the register busy)
template <bool SPLIT, int WAVES, bool PREFETCH>
__global__ __launch_bounds__(256, WAVES) void k_roundeval9(const uint32_t *__restrict__ a_hi, const uint32_t *__restrict__ a_lo,
                                                      const uint32_t *__restrict__ b_hi, const uint32_t *__restrict__ b_lo,
                                                      uint64_t n, f128 *out, fin_fuse fz)
{
	__shared__ uint4 tile[4][kWaveQ];

	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned g = lane / 9, c = lane - g * 9;
	const bool live = lane < 63;
	const bool loader = live && c < 8;
	uint4 *wt = tile[wave];

	// zero block (read by combination slots a lane does not use)
	if (lane < kBlkQ)
		wt[kZeroBlk * kBlkQ + lane] = uint4{0, 0, 0, 0};

	// which words this lane loads: c = 0..3 -> a word c ; c = 4..7 -> b word c-4
	const unsigned w = c & 3;
	const uint32_t *p_hi = (c & 4) ? b_hi : a_hi;
	const uint32_t *p_lo = (c & 4) ? b_lo : a_lo;
	// combination mask over limbs 0..3
	unsigned mask;
	switch (c) {
	case 0: mask = 1; break;
	case 1: mask = 2; break;
	case 2: mask = 3; break;
	case 3: mask = 4; break;
	case 4: mask = 8; break;
	case 5: mask = 12; break;
	case 6: mask = 5; break;
	case 7: mask = 10; break;
	default: mask = 15; break;
	}
	if (!live) mask = 0;
	// LDS offsets (in uint4) of the four a-slots and four b-slots this lane reads
	unsigned off_a[4], off_b[4];
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const bool use = (mask >> s) & 1;
		off_a[s] = (use ? (unsigned)(s * kGroups + g) : (unsigned)kZeroBlk) * kBlkQ;
		off_b[s] = (use ? (unsigned)((4 + s) * kGroups + g) : (unsigned)kZeroBlk) * kBlkQ;
	}
	const unsigned off_w = (loader ? (c * kGroups + g) : 0u) * kBlkQ; // where a loader writes its limb

	uint32_t acc[32];
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;

	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + wave;
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;

	// rows of one batch: point j of this group is b*112 + 7*j + g, i.e. byte offset j*112 from the
	// group's first row -> immediate offsets on one base pointer.  Non-loader lanes (c == 8, lane 63)
	// load harmless in-range words and never publish them.  Only the last batch can be ragged.
	const uint32_t *q_hi = p_hi + w, *q_lo = p_lo + w;
	const unsigned g_ld = live ? g : 0;
	auto load_rows = [&](uint64_t b, uint32_t (&dst)[32]) {
		const uint64_t base = b * kBatch + g_ld;
		if ((b + 1) * kBatch <= n) {
			const uint32_t *h = q_hi + (base << 2), *l = q_lo + (base << 2);
#pragma unroll
			for (int j = 0; j < 16; j++) {
				dst[j] = h[28 * j];
				dst[16 + j] = l[28 * j];
			}
		} else {
#pragma unroll
			for (int j = 0; j < 16; j++) {
				const uint64_t e = base + 7 * j;
				const bool ok = e < n;
				const uint64_t idx = ok ? (e << 2) : 0;
				const uint32_t vh = q_hi[idx], vl = q_lo[idx];
				dst[j] = ok ? vh : 0u;
				dst[16 + j] = ok ? vl : 0u;
			}
		}
	};
	uint32_t rn[32];
	if (PREFETCH && wave_global < n_batches)
		load_rows(wave_global, rn);
	for (uint64_t b = wave_global; b < n_batches; b += n_waves) {
		uint32_t r[32];
		if (PREFETCH) {
#pragma unroll
			for (int j = 0; j < 32; j++)
				r[j] = rn[j];
		} else {
			load_rows(b, r);
		}
		// "lo + hi" on the raw rows (the transpose is linear): rows 16..31 become lo ^ hi
		if (!SPLIT) {
#pragma unroll
			for (int j = 0; j < 16; j++)
				r[16 + j] ^= r[j];
		}
		transpose32(r);
		// publish this limb to the wave's LDS tile
		if (loader) {
#pragma unroll
			for (int q = 0; q < 8; q++)
				wt[off_w + q] = uint4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// software prefetch: the next batch's rows fly while this batch is multiplied
		if (PREFETCH && b + n_waves < n_batches)
			load_rows(b + n_waves, rn);
		// gather the combination: A = XOR of a-limbs in `mask`, B likewise
		uint32_t A[32], B[32];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint4 x0 = wt[off_a[0] + q], x1 = wt[off_a[1] + q], x2 = wt[off_a[2] + q], x3 = wt[off_a[3] + q];
			const uint4 y0 = wt[off_b[0] + q], y1 = wt[off_b[1] + q], y2 = wt[off_b[2] + q], y3 = wt[off_b[3] + q];
			A[4 * q] = xor3(x0.x, x1.x, x2.x) ^ x3.x;
			A[4 * q + 1] = xor3(x0.y, x1.y, x2.y) ^ x3.y;
			A[4 * q + 2] = xor3(x0.z, x1.z, x2.z) ^ x3.z;
			A[4 * q + 3] = xor3(x0.w, x1.w, x2.w) ^ x3.w;
			B[4 * q] = xor3(y0.x, y1.x, y2.x) ^ y3.x;
			B[4 * q + 1] = xor3(y0.y, y1.y, y2.y) ^ y3.y;
			B[4 * q + 2] = xor3(y0.z, y1.z, y2.z) ^ y3.z;
			B[4 * q + 3] = xor3(y0.w, y1.w, y2.w) ^ y3.w;
			if (q & 1)
				__builtin_amdgcn_sched_barrier(0); // at most two quads' 16 reads in flight (register pressure)
		}
		// the tile may be overwritten by the next batch only after these reads: DS ops of a wave
		// execute in order, the fences keep the compiler from reordering across them
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		uint32_t P[32];
		bs_mul<5>(A, B, P);
#pragma unroll
		for (int p = 0; p < 32; p++)
			acc[p] ^= P[p];
	}

	re9::tail(acc, live, c, g, wave, lane, out, fz, fz.args.seq);
}

static unsigned grid9(uint64_t n, int n_cu, int waves)
{
	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	uint64_t blocks = (n_batches + 3) / 4;
	if (blocks < 1) blocks = 1;
	const uint64_t cap = (uint64_t)n_cu * waves; // `waves` 256-thread blocks per CU = waves per SIMD
	return (unsigned)(blocks < cap ? blocks : cap);
}

template <bool SPLIT>
static hipError_t launch9(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo,
                          uint64_t n, f128 *d_out, const fin_fuse *fuse)
{
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	if (n == 0 && fuse) return hipErrorNotSupported; // nothing to launch: the caller finalizes separately
	if (n == 0) return hipSuccess;
	const uint32_t *p0 = (const uint32_t *)a_hi, *p1 = (const uint32_t *)a_lo, *p2 = (const uint32_t *)b_hi, *p3 = (const uint32_t *)b_lo;
	// 2 waves per SIMD with register prefetch measured fastest on MI355X (3 waves/SIMD spills to
	// scratch: 0.36-0.70 ms vs 0.235 ms at n = 24; no prefetch: 0.277 ms) -- profiles/r01/.
	hipLaunchKernelGGL((k_roundeval9<SPLIT, 2, true>), dim3(grid9(n, n_cu, 2)), dim3(256), 0, s, p0, p1, p2, p3, n, d_out, fz);
	return hipGetLastError();
}

// d_out[0] ^= sum_i a_hi[i]*b_hi[i] ; d_out[1] ^= sum_i (a_lo[i]^a_hi[i])*(b_lo[i]^b_hi[i])
hipError_t launch_roundeval9_pair(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi,
                                  const void *b_lo, uint64_t n, f128 *d_out, const fin_fuse *fuse)
{
	return launch9<false>(s, n_cu, a_hi, a_lo, b_hi, b_lo, n, d_out, fuse);
}

// d_out[0] ^= sum_{i<n} a[i]*b[i] ; d_out[1] ^= sum_{i<n} a[i+split]*b[i+split]
hipError_t launch_roundeval9_split(hipStream_t s, int n_cu, const void *a, const void *b, uint64_t n, uint64_t split_off,
                                   f128 *d_out)
{
	const char *a2 = (const char *)a + split_off * 16, *b2 = (const char *)b + split_off * 16;
	return launch9<true>(s, n_cu, a, a2, b, b2, n, d_out, nullptr);
}

} // namespace bn
