// binius_amd/csrc/kernels_foldeval9.hip -- fold of round r and round evaluation of round r+1 in ONE
// pass over the data, for the bivariate product a*b:
//
//   a'[i] = a[i] + z*(a[i + N/2] - a[i])        i < N/2        (extrapolate_line, layer.rs:421)
//   S_1   = sum_{j < N/4} a'[j + N/4] * b'[j + N/4]
//   S_inf = sum_{j < N/4} (a'[j] + a'[j + N/4]) * (b'[j] + b'[j + N/4])
//                                                (v3/bivariate_product.rs:217-228 then :303-408)
//
// The reference's prover issues the fold of round r and the evaluation of round r+1 as two
// back-to-back ComputeLayer calls with no host decision in between; the ABI defers the fold
// (abi.cpp, "pending fold") and, when the next kernel launch evaluates exactly the folded arrays,
// runs this kernel instead of two.  The folded values are written back in place (they are the next
// round's input) and consumed from LDS for the evaluation: the evaluation's own HBM read
// (16*m*N/2 bytes) and one kernel launch + inter-kernel gap per round disappear.
//
// Per wave-batch of 112 points: 448 folded elements (2 arrays x {lo', hi'} x 112), 7 per lane.
// Lane L owns elements L, L+64, ...: 16-byte coalesced loads of x0 = X[e], x1 = X[e + N/2], the
// constant multiplication by z through the LDS nibble tables (ctable.hpp), a coalesced 16-byte
// store, and a copy into the wave's LDS tile, from which the 56 loader lanes pick their 32-bit word
// columns exactly as k_roundeval9 picks them from global memory.  The staging copy aliases the
// exchange tile (it is dead once the rows are in registers).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ctable.hpp"
#include "re9.hpp"

namespace bn {

using namespace re9;

namespace {
constexpr int kArrQ = 232;          // staging uint4 per array: 2 x 112 rows + 8 pad (arrays 32 banks apart)
constexpr int kSlots = 8;           // fold slots per lane and batch: 4 quadrants x 2 (112 = 64 + 48 points)
constexpr int kPre = 4;             // slots prefetched across the multiply (array a); the rest load in the fold phase
static_assert(2 * kArrQ <= kZeroBlk * kBlkQ, "staging must not touch the tile's zero block");
} // namespace

// Fold slot t of a lane: quadrant q = t/2 = (array, half) and point lane + 64*(t&1) of the batch; the
// second slot of a quadrant is live in lanes 0..47 only.  Everything but `lane` is compile-time or
// wave-uniform, so the addresses are one scalar base per quadrant + (lane*16 + immediate): no
// per-lane address registers survive into the multiply (the kernel sits at the 256-VGPR limit of
// two waves per SIMD).
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_foldeval9(foldeval_args fa, uint64_t n_in, f128 z, f128 *out, fin_fuse fz)
{
	__shared__ uint4 tile[4][kWaveQ];
	__shared__ ctable_smem tab;
	ctable_build(tab, z);

	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned g = lane / 9, c = lane - g * 9;
	const bool live = lane < 63;
	const bool loader = live && c < 8;
	uint4 *wt = tile[wave];
	const uint32_t *wtw = reinterpret_cast<const uint32_t *>(wt);
	// zero block (read by combination slots a lane does not use)
	if (lane < kBlkQ)
		wt[kZeroBlk * kBlkQ + lane] = uint4{0, 0, 0, 0};

	const uint64_t n = n_in >> 2;     // evaluation points of the next round

	unsigned mask = live ? combo_mask(c) : 0u;
	unsigned off_a[4], off_b[4];
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const bool use = (mask >> s) & 1;
		off_a[s] = (use ? (unsigned)(s * kGroups + g) : (unsigned)kZeroBlk) * kBlkQ;
		off_b[s] = (use ? (unsigned)((4 + s) * kGroups + g) : (unsigned)kZeroBlk) * kBlkQ;
	}
	const unsigned off_w = (loader ? (c * kGroups + g) : 0u) * kBlkQ;
	// staging word offsets of this loader's column: array c>>2, word c&3, rows 7*j + g
	const unsigned g_ld = live ? g : 0;
	const unsigned st_lo = (((c >> 2) & 1) * kArrQ + g_ld) * 4 + (c & 3); // lo' rows (first 112)
	const unsigned st_hi = st_lo + kBatch * 4;                            // hi' rows

	uint32_t acc[32];
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;

	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + wave;
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;

	uint4 x0[kSlots], x1[kSlots];
	// quadrant base of slot t for batch point p0 (wave-uniform)
	auto qoff = [&](int t, uint64_t p0) -> uint64_t { return (((t >> 1) & 1) ? n : 0) + p0; };
	auto load_raw = [&](uint64_t b, auto t0c, auto t1c) {
		constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
		const uint64_t p0 = b * kBatch;
		const uint64_t left = n - p0; // points left from p0 (>= 1)
#pragma unroll
		for (int t = T0; t < T1; t++) {
			const unsigned pt = lane + 64 * (t & 1);
			const uint4 *q0 = (const uint4 *)fa.x0[t >> 2] + qoff(t, p0), *q1 = (const uint4 *)fa.x1[t >> 2] + qoff(t, p0);
			uint4 v0{0, 0, 0, 0}, v1{0, 0, 0, 0};
			if (pt < kBatch && pt < left) {
				v0 = q0[pt];
				v1 = q1[pt];
			}
			x0[t] = v0;
			x1[t] = v1;
		}
	};
	using c0 = std::integral_constant<int, 0>;
	using cP = std::integral_constant<int, kPre>;
	using cN = std::integral_constant<int, kSlots>;
	if (wave_global < n_batches)
		load_raw(wave_global, c0{}, cP{});
	for (uint64_t b = wave_global; b < n_batches; b += n_waves) {
		const uint64_t p0 = b * kBatch;
		const uint64_t left = n - p0;
		load_raw(b, cP{}, cN{});
		// ---- fold: f = x0 + z * (x0 + x1); back to HBM (next round's input) and into the staging tile
#pragma unroll
		for (int t = 0; t < kSlots; t++) {
			const unsigned pt = lane + 64 * (t & 1);
			const uint4 f = xor4(x0[t], ctable_mul(tab, xor4(x0[t], x1[t])));
			if (pt < kBatch) {
				if (pt < left)
					((uint4 *)fa.out[t >> 2] + qoff(t, p0))[pt] = f;
				wt[(t >> 2) * kArrQ + ((t >> 1) & 1) * kBatch + pt] = f; // points past the end carry zeros
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// ---- loader lanes pick their word column: rows 0..15 = hi', rows 16..31 = lo'
		uint32_t r[32];
#pragma unroll
		for (int j = 0; j < 16; j++) {
			r[j] = wtw[st_hi + 28 * j];
			r[16 + j] = wtw[st_lo + 28 * j];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// next batch's raw elements fly while this batch is transposed and multiplied
		if (b + n_waves < n_batches)
			load_raw(b + n_waves, c0{}, cP{});
#pragma unroll
		for (int j = 0; j < 16; j++)
			r[16 + j] ^= r[j];
		transpose32(r);
		if (loader) {
#pragma unroll
			for (int q = 0; q < 8; q++)
				wt[off_w + q] = uint4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		uint32_t A[32], B[32];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint4 a0 = wt[off_a[0] + q], a1 = wt[off_a[1] + q], a2 = wt[off_a[2] + q], a3 = wt[off_a[3] + q];
			const uint4 y0 = wt[off_b[0] + q], y1 = wt[off_b[1] + q], y2 = wt[off_b[2] + q], y3 = wt[off_b[3] + q];
			A[4 * q] = xor3(a0.x, a1.x, a2.x) ^ a3.x;
			A[4 * q + 1] = xor3(a0.y, a1.y, a2.y) ^ a3.y;
			A[4 * q + 2] = xor3(a0.z, a1.z, a2.z) ^ a3.z;
			A[4 * q + 3] = xor3(a0.w, a1.w, a2.w) ^ a3.w;
			B[4 * q] = xor3(y0.x, y1.x, y2.x) ^ y3.x;
			B[4 * q + 1] = xor3(y0.y, y1.y, y2.y) ^ y3.y;
			B[4 * q + 2] = xor3(y0.z, y1.z, y2.z) ^ y3.z;
			B[4 * q + 3] = xor3(y0.w, y1.w, y2.w) ^ y3.w;
			if (q & 1)
				__builtin_amdgcn_sched_barrier(0);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		uint32_t P[32];
		bs_mul<5>(A, B, P);
#pragma unroll
		for (int p = 0; p < 32; p++)
			acc[p] ^= P[p];
	}

	re9::tail(acc, live, c, g, wave, lane, out, fz);
}

// For both arrays j: out_j[i] = x0_j[i] + z * (x1_j[i] - x0_j[i]), i < n_in/2 (out_j may be x0_j), and
// accumulate the next round's (S_1, S_inf) of out_0 * out_1 into d_out[0], d_out[1].  n_in >= 4.
hipError_t launch_foldeval9(hipStream_t s, int n_cu, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse *fuse)
{
	if (n_in < 4 || (n_in & 3)) return hipErrorNotSupported;
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	const uint64_t n = n_in >> 2;
	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	uint64_t blocks = (n_batches + 3) / 4;
	const uint64_t cap = (uint64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL((k_foldeval9<2>), dim3((unsigned)blocks), dim3(256), 0, s, fa, n_in, z, d_out, fz);
	return hipGetLastError();
}

} // namespace bn
