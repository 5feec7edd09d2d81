// binius_amd/csrc/kernels_roundeval9_eq.hip -- round evaluation of the MLE-check composition
// a * b * eq_ind (crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:391-520):
//   S_1   = sum_i a_hi[i] * b_hi[i] * eq[i]
//   S_inf = sum_i (a_lo[i] + a_hi[i]) * (b_lo[i] + b_hi[i]) * eq[i]
// (the eq indicator chunk is the same at both evaluation points).
//
// Two chained products per point.  The first one (a * b) must be a proper field element before it
// can be multiplied again, so the lazy trick of kernels_roundeval9.hip applies only to the SECOND
// product: per batch the 9 limb-combination products of a * b are exchanged through LDS, four lanes
// of every group rebuild the four 32-bit limbs of the product in the bit-sliced domain (the scheme of
// kernels_mul9.hip, with the alpha multiples published by the PRODUCING lanes so that a builder lane
// only XORs <= 12 blocks, four planes at a time), the limbs go back into the exchange tile as the
// "a" operand, the transposed eq words as the "b" operand, and the
// second product is accumulated lazily exactly like the bivariate kernel does.  Both evaluation
// points ride in the two 16-bit halves of every plane register throughout.
// ~3500 VALU per 112 points instead of the ~8000 (with spills) of the generic 128-plane kernel.
#include <hip/hip_runtime.h>

#include "re9.hpp"

namespace bn {

using namespace re9;

namespace {
// block types: 0..8 = partial products p_c; 9..13 = alpha * p_{1,3,4,5,7}; 14 = alpha^2 * p_4
constexpr int kTypes = 15;
constexpr int kPBlocks = kTypes * kGroups; // 105 blocks (the 56 limb blocks alias the first ones)
constexpr int kZeroP = kPBlocks;           // zero block
constexpr int kWaveQP = (kPBlocks + 1) * kBlkQ;
// limb L of a * b = XOR of these block types (derivation: kernels_mul9.hip header)
//   R0 = p0+p1+p3+p4                R1 = p0..p5 + a(p1) + a(p4)
//   R2 = p0+p1+p5+p6+p7 + a(p4)     R3 = p0+p1+p2+p5+p6+p7+p8 + a(p1)+a(p3)+a(p5)+a(p7) + aa(p4)
constexpr int kNone = -1;
__device__ constexpr int kRebuild[4][12] = {
    {0, 1, 3, 4, kNone, kNone, kNone, kNone, kNone, kNone, kNone, kNone},
    {0, 1, 2, 3, 4, 5, 9, 11, kNone, kNone, kNone, kNone},
    {0, 1, 5, 6, 7, 11, kNone, kNone, kNone, kNone, kNone, kNone},
    {0, 1, 2, 5, 6, 7, 8, 9, 10, 12, 13, 14},
};
} // namespace

__global__ __launch_bounds__(256, 2) void k_roundeval9_eq(const uint32_t *__restrict__ a_hi, const uint32_t *__restrict__ a_lo,
                                                          const uint32_t *__restrict__ b_hi, const uint32_t *__restrict__ b_lo,
                                                          const uint32_t *__restrict__ eq, uint64_t n, f128 *out, fin_fuse fz)
{
	__shared__ uint4 tile[4][kWaveQP];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned g = lane / 9, c = lane - g * 9;
	const bool live = lane < 63;
	const bool loader = live && c < 8;
	const bool builder = live && c < 4;
	uint4 *wt = tile[wave];
	// stores go through a volatile 128-bit LDS pointer: otherwise some of the 8-chunk block stores are
	// re-split into ds_write2_b32, whose 32-bank mapping conflicts 4-way on the 144-byte block stride
	typedef unsigned int u4v __attribute__((ext_vector_type(4)));
	typedef __attribute__((address_space(3))) volatile u4v lds_vu4;
	lds_vu4 *wv = (lds_vu4 *)(__attribute__((address_space(3))) void *)wt;
	if (lane < kBlkQ)
		wt[kZeroP * kBlkQ + lane] = uint4{0, 0, 0, 0};

	const unsigned w = c & 3;
	const uint32_t *p_hi = (c & 4) ? b_hi : a_hi;
	const uint32_t *p_lo = (c & 4) ? b_lo : a_lo;
	const unsigned mask = live ? combo_mask(c) : 0u;
	unsigned off_a[4], off_b[4];
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const bool use = (mask >> s) & 1;
		off_a[s] = (use ? (unsigned)(s * kGroups + g) : (unsigned)kZeroP) * kBlkQ;
		off_b[s] = (use ? (unsigned)((4 + s) * kGroups + g) : (unsigned)kZeroP) * kBlkQ;
	}
	const unsigned g_ld = live ? g : 0;
	const unsigned off_w = (loader ? (c * kGroups + g_ld) : 0u) * kBlkQ;               // limb slot of a loader / builder lane
	const unsigned off_pp = (live ? (c * kGroups + g) : (unsigned)kZeroP) * kBlkQ;      // partial-product slot
	// alpha-multiple slot of this lane's product (types 9..13), if anybody needs it
	const int atype = c == 1 ? 9 : c == 3 ? 10 : c == 4 ? 11 : c == 5 ? 12 : c == 7 ? 13 : -1;
	const unsigned off_ap = ((live && atype >= 0) ? (unsigned)(atype * kGroups) + g : (unsigned)kZeroP) * kBlkQ;
	const unsigned off_aap = ((live && c == 4) ? (unsigned)(14 * kGroups) + g : (unsigned)kZeroP) * kBlkQ;

	uint32_t acc[32];
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;

	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + wave;
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;
	const uint32_t *q_hi = p_hi + w, *q_lo = p_lo + w, *q_eq = eq + w;

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
	auto load_eq = [&](uint64_t b, uint32_t (&dst)[16]) {
		const uint64_t base = b * kBatch + g_ld;
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const uint64_t e = base + 7 * j;
			const bool ok = e < n;
			const uint32_t v = q_eq[ok ? (e << 2) : 0];
			dst[j] = ok ? v : 0u;
		}
	};
	auto gather = [&](uint32_t (&A)[32], uint32_t (&B)[32]) {
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
				__builtin_amdgcn_sched_barrier(0);
		}
	};
	auto wave_sync = [&]() {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};

	// The next batch's rows are prefetched only across the SECOND product (issued after the rebuild):
	// 32 more live registers across the first product and the rebuild as well push the kernel far into
	// scratch (420 spilled VGPRs, measured).
	// No register prefetch of the next batch's rows: 32 more live registers anywhere in this loop body
	// push the kernel far into scratch (300-420 spilled VGPRs, measured in three placements), and one
	// wave per SIMD with AGPR spills is slower (1.02 vs 0.83 ms at n = 24).  The second wave of the
	// SIMD covers the load latency.
	for (uint64_t b = wave_global; b < n_batches; b += n_waves) {
		uint32_t r[32];
		load_rows(b, r);
#pragma unroll
		for (int j = 0; j < 16; j++)
			r[16 + j] ^= r[j];
		transpose32(r);
		if (loader) {
#pragma unroll
			for (int q = 0; q < 8; q++)
				wv[off_w + q] = u4v{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
		}
		wave_sync();
		// the eq rows of this batch fly during the first product
		uint32_t er[16];
		load_eq(b, er);
		{
			uint32_t A[32], B[32], P[32];
			gather(A, B);
			wave_sync();
			bs_mul<5>(A, B, P);
			// publish the partial product of a * b and the alpha multiples others need (the limb tile is
			// dead now: same LDS region).  Lanes without a consumer write the zero block with zeros?  No:
			// they write their values into the zero block only if those are zero -- so they skip instead.
#pragma unroll
			for (int q = 0; q < 8; q++)
				wv[off_pp + q] = u4v{P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]};
			bs_mul_alpha<5>(P, A); // A = alpha * P
			if (atype >= 0 && live) {
#pragma unroll
				for (int q = 0; q < 8; q++)
					wv[off_ap + q] = u4v{A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};
			}
			bs_mul_alpha<5>(A, B); // B = alpha^2 * P
			if (c == 4 && live) {
#pragma unroll
				for (int q = 0; q < 8; q++)
					wv[off_aap + q] = u4v{B[4 * q], B[4 * q + 1], B[4 * q + 2], B[4 * q + 3]};
			}
		}
		wave_sync();
		// ---- rebuild limb c of a * b (lanes c < 4): XOR of <= 12 blocks, four planes at a time
		{
			unsigned cb = builder ? c : 0u, gl = g;
			asm volatile("" : "+v"(cb), "+v"(gl)); // (keeps the 12 offsets from living across the products)
			unsigned offs[12];
#pragma unroll
			for (int k = 0; k < 12; k++) {
				const int ty = cb == 0 ? kRebuild[0][k] : cb == 1 ? kRebuild[1][k] : cb == 2 ? kRebuild[2][k] : kRebuild[3][k];
				offs[k] = ((builder && ty >= 0) ? (unsigned)(ty * kGroups) + gl : (unsigned)kZeroP) * kBlkQ;
			}
#pragma unroll
			for (int q = 0; q < 8; q++) {
				uint4 x{0, 0, 0, 0};
#pragma unroll
				for (int k = 0; k < 12; k++) {
					const uint4 t = wt[offs[k] + q];
					x.x ^= t.x; x.y ^= t.y; x.z ^= t.z; x.w ^= t.w;
				}
				r[4 * q] = x.x; r[4 * q + 1] = x.y; r[4 * q + 2] = x.z; r[4 * q + 3] = x.w;
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		wave_sync(); // every partial product has been read: the region may become limb blocks again
		// ---- second product: operand "a" = the rebuilt limbs (lanes c < 4), operand "b" = the eq words
		// (lanes c = 4..7), the same 16 rows for both evaluation points
		{
			uint32_t e[32];
#pragma unroll
			for (int j = 0; j < 16; j++) {
				e[j] = er[j];
				e[16 + j] = er[j];
			}
			transpose32(e);
			if (loader) {
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const uint4 vr = uint4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
					const uint4 ve = uint4{e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]};
					wt[off_w + q] = builder ? vr : ve;
				}
			}
		}
		wave_sync();
		{
			uint32_t A[32], B[32], P[32];
			gather(A, B);
			wave_sync();
			bs_mul<5>(A, B, P);
#pragma unroll
			for (int p = 0; p < 32; p++)
				acc[p] ^= P[p];
		}
	}
	re9::tail<4>(acc, live, c, g, wave, lane, out, fz, fz.args.seq);
}

// d_out[0] ^= sum_i a_hi*b_hi*eq ; d_out[1] ^= sum_i (a_lo+a_hi)*(b_lo+b_hi)*eq
hipError_t launch_roundeval9_eq(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo,
                                const void *eq, uint64_t n, f128 *d_out, const fin_fuse *fuse)
{
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	if (n == 0) return fuse ? hipErrorNotSupported : hipSuccess;
	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	uint64_t blocks = (n_batches + 3) / 4;
	const uint64_t cap = (uint64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(k_roundeval9_eq, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t *)a_hi, (const uint32_t *)a_lo,
	                   (const uint32_t *)b_hi, (const uint32_t *)b_lo, (const uint32_t *)eq, n, d_out, fz);
	return hipGetLastError();
}

} // namespace bn
