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

#include "arm.hpp"
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
// per-lane constants of the evaluation stage
struct eval_layout {
	unsigned off_a[4], off_b[4]; // LDS offsets (uint4) of the a- and b-limb blocks this lane combines
	unsigned off_w;              // where a loader lane publishes its transposed limb
	unsigned st_lo, st_hi;       // staging word offsets of this loader's column (lo' rows, hi' rows)
	bool loader;
};

__device__ __forceinline__ eval_layout make_eval_layout(unsigned lane)
{
	const unsigned g = lane / 9, c = lane - g * 9;
	const bool live = lane < 63;
	eval_layout lay;
	lay.loader = live && c < 8;
	const unsigned mask = live ? combo_mask(c) : 0u;
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const bool use = (mask >> s) & 1;
		lay.off_a[s] = (use ? (unsigned)(s * kGroups + g) : (unsigned)kZeroBlk) * kBlkQ;
		lay.off_b[s] = (use ? (unsigned)((4 + s) * kGroups + g) : (unsigned)kZeroBlk) * kBlkQ;
	}
	lay.off_w = (lay.loader ? (c * kGroups + g) : 0u) * kBlkQ;
	// staging word offsets of this loader's column: array c>>2, word c&3, rows 7*j + g
	const unsigned g_ld = live ? g : 0;
	lay.st_lo = (((c >> 2) & 1) * kArrQ + g_ld) * 4 + (c & 3); // lo' rows (first 112)
	lay.st_hi = lay.st_lo + kBatch * 4;                        // hi' rows
	return lay;
}

// Evaluation stage of one batch: the folded elements are staged in the wave's tile wt (array-major,
// lo' rows then hi' rows); accumulates this lane's limb-combination product into acc.
// after_read() runs once the rows are in registers (the staging area is dead from then on).
template <class F>
__device__ __forceinline__ void eval_staged(uint4 *wt, const eval_layout &lay, uint32_t (&acc)[32], F &&after_read)
{
	const uint32_t *wtw = reinterpret_cast<const uint32_t *>(wt);
	// ---- lay.loader lanes pick their word column: rows 0..15 = hi', rows 16..31 = lo'
	uint32_t r[32];
#pragma unroll
	for (int j = 0; j < 16; j++) {
		r[j] = wtw[lay.st_hi + 28 * j];
		r[16 + j] = wtw[lay.st_lo + 28 * j];
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	after_read();
#pragma unroll
	for (int j = 0; j < 16; j++)
		r[16 + j] ^= r[j];
	transpose32(r);
	if (lay.loader) {
#pragma unroll
		for (int q = 0; q < 8; q++)
			wt[lay.off_w + q] = uint4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	uint32_t A[32], B[32];
#pragma unroll
	for (int q = 0; q < 8; q++) {
		const uint4 a0 = wt[lay.off_a[0] + q], a1 = wt[lay.off_a[1] + q], a2 = wt[lay.off_a[2] + q], a3 = wt[lay.off_a[3] + q];
		const uint4 y0 = wt[lay.off_b[0] + q], y1 = wt[lay.off_b[1] + q], y2 = wt[lay.off_b[2] + q], y3 = wt[lay.off_b[3] + q];
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
	BN_TS(4);
	uint32_t P[32];
	bs_mul<5, false>(A, B, P); // the Karatsuba form: see bitslice.hpp (register pressure)
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] ^= P[p];
}

// One round for one wave: folds and evaluates batches wave_global, wave_global + n_waves, ... and
// leaves the wave's 32 accumulator planes in acc.  wt = this wave's LDS tile (kWaveQ uint4).
template <bool PREFETCH = true>
__device__ __forceinline__ void foldeval_wave(const foldeval_args &fa, uint64_t n_in, const ctable_smem &tab, uint4 *wt,
                                              uint64_t wave_global, uint64_t n_waves, uint32_t (&acc)[32], const ctable_smem *tab_hs = nullptr)
{
	const unsigned lane = threadIdx.x & 63;
	const eval_layout lay = make_eval_layout(lane);
	// zero block (read by combination slots a lane does not use)
	if (lane < kBlkQ)
		wt[kZeroBlk * kBlkQ + lane] = uint4{0, 0, 0, 0};

	const uint64_t n = n_in >> 2;     // evaluation points of the next round

#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;

	const uint64_t n_batches = (n + kBatch - 1) / kBatch;

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
	if (PREFETCH && wave_global < n_batches)
		load_raw(wave_global, c0{}, cP{});
	for (uint64_t b = wave_global; b < n_batches; b += n_waves) {
		const uint64_t p0 = b * kBatch;
		const uint64_t left = n - p0;
		if (PREFETCH) {
			load_raw(b, cP{}, cN{});
		} else {
			load_raw(b, c0{}, cN{}); // resident tail: a wave sees one or two batches, nothing to overlap
		}
		// ---- fold: f = x0 + z * (x0 + x1); back to HBM (next round's input) and into the staging tile
#pragma unroll
		for (int t = 0; t < kSlots; t++) {
			const unsigned pt = lane + 64 * (t & 1);
			uint4 f = xor4(x0[t], ctable_mul<8>(tab, xor4(x0[t], x1[t])));
			// slot t: array t >> 2, half (t >> 1) & 1 -- the scaled fold multiplies the upper half of a marked array
			if (tab_hs && ((t >> 1) & 1) && ((fa.scale_mask >> (t >> 2)) & 1)) f = ctable_mul<8>(*tab_hs, f);
			if (pt < kBatch) {
				if (pt < left)
					((uint4 *)fa.out[t >> 2] + qoff(t, p0))[pt] = f;
				wt[(t >> 2) * kArrQ + ((t >> 1) & 1) * kBatch + pt] = f; // points past the end carry zeros
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		eval_staged(wt, lay, acc, [&]() {
			// next batch's raw elements fly while this batch is transposed and multiplied
			if (PREFETCH && b + n_waves < n_batches)
				load_raw(b + n_waves, c0{}, cP{});
		});
	}

}

template <int WAVES, bool SCALED = false>
__global__ __launch_bounds__(256, WAVES) void k_foldeval9(foldeval_args fa, uint64_t n_in, f128 z, f128 *out, fin_fuse fz, arm_args arm)
{
	__shared__ uint4 tile[4][kWaveQ];
	__shared__ ctable_smem tab;
	__shared__ ctable_opt<SCALED> tab_hs;
	__shared__ fin_cache fcache;
	const uint64_t seq = fz.args.seq;
	const fin_pref fpre = fin_prefetch(fz); // the finalize arguments wait in LDS for the tail (finalize.hpp)
	if (arm.h_cmd) { // (uniform) armed launch: the challenge arrives through the command block (arm.hpp)
		f128 hs_in;
		if (!arm_wait(arm, z, hs_in)) return;
		fa.hi_scale = hs_in;
	}
	const ctable_smem *hs = nullptr;
	if constexpr (SCALED) {
		// both tables side by side (two halves of the workgroup): a build is one dependent chain whatever the number of threads
		const unsigned half = blockDim.x >> 1, grp = threadIdx.x >= half ? 1u : 0u;
		ctable_build_group(grp ? tab_hs.get() : tab, grp ? fa.hi_scale : z, threadIdx.x - grp * half, half);
		hs = &tab_hs.get();
	} else {
		ctable_build(tab, z);
	}
	fin_commit(fz, fpre, fcache);
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned g = lane / 9, c = lane - g * 9;
	uint32_t acc[32];
	foldeval_wave(fa, n_in, tab, tile[wave], (uint64_t)blockIdx.x * 4 + wave, (uint64_t)gridDim.x * 4, acc, hs);
	re9::tail<4>(acc, lane < 63, c, g, wave, lane, out, fz, seq, &fcache);
}

// Latency-shaped variant for small rounds (every batch gets its own workgroup: n_batches <= 2 per CU).
// A lone wave issues one VALU instruction per ~4.7 cycles, so the 8 serial constant multiplications
// of a batch (1600 instructions) cost 3 us on one wave; here the four waves of the workgroup fold one
// quadrant each (2 slots), with the loads in flight while the nibble tables are built, and wave 0
// evaluates the batch.
template <bool SCALED>
__global__ __launch_bounds__(256, 2) void k_foldeval9_small(foldeval_args fa, uint64_t n_in, f128 z, f128 *out, fin_fuse fz, arm_args arm)
{
	__shared__ uint4 tile[kWaveQ];
	__shared__ ctable_smem tab;
	__shared__ ctable_opt<SCALED> tab_hs;
	__shared__ fin_cache fcache;
	BN_TS(0);
	const uint64_t seq = fz.args.seq;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned g = lane / 9, c = lane - g * 9;
	const uint64_t n = n_in >> 2;
	const uint64_t p0 = (uint64_t)blockIdx.x * kBatch;
	const uint64_t left = n - p0;
	// this wave's quadrant: array wave>>1, half wave&1; two slots: points lane, lane + 64
	const uint64_t qo = ((wave & 1) ? n : 0) + p0;
	const uint4 *q0 = (const uint4 *)fa.x0[wave >> 1] + qo, *q1 = (const uint4 *)fa.x1[wave >> 1] + qo;
	uint4 *qd = (uint4 *)fa.out[wave >> 1] + qo;
	uint4 x0[2], x1[2];
#pragma unroll
	for (int sub = 0; sub < 2; sub++) {
		const unsigned pt = lane + 64 * sub;
		uint4 v0{0, 0, 0, 0}, v1{0, 0, 0, 0};
		if (pt < kBatch && pt < left) {
			v0 = q0[pt];
			v1 = q1[pt];
		}
		x0[sub] = v0;
		x1[sub] = v1;
	}
	const fin_pref fpre = fin_prefetch(fz); // the finalize arguments: in flight with the data
	if (arm.h_cmd) { // (uniform) armed launch: data and arguments are on their way, the challenge is what is missing (arm.hpp)
		f128 hs_in;
		if (!arm_wait(arm, z, hs_in)) return;
		fa.hi_scale = hs_in;
	}
	BN_TS(1);
	if constexpr (SCALED) { // (both tables side by side, see above; the loads above are in flight meanwhile)
		const unsigned half = blockDim.x >> 1, grp = threadIdx.x >= half ? 1u : 0u;
		ctable_build_group(grp ? tab_hs.get() : tab, grp ? fa.hi_scale : z, threadIdx.x - grp * half, half);
	} else {
		ctable_build(tab, z); // the loads above are in flight meanwhile
	}
	BN_TS(2);
	const bool scaled_quadrant = SCALED && (wave & 1) && ((fa.scale_mask >> (wave >> 1)) & 1); // (wave-uniform)
	if (threadIdx.x < kBlkQ)
		tile[kZeroBlk * kBlkQ + threadIdx.x] = uint4{0, 0, 0, 0};
#pragma unroll
	for (int sub = 0; sub < 2; sub++) {
		const unsigned pt = lane + 64 * sub;
		if (sub == 1 && left <= 64) break; // (uniform) nothing in the second slot
		uint4 f = xor4(x0[sub], ctable_mul(tab, xor4(x0[sub], x1[sub])));
		if constexpr (SCALED)
			if (scaled_quadrant) f = ctable_mul(tab_hs.get(), f);
		if (pt < kBatch) {
			if (pt < left) qd[pt] = f;
			tile[(wave >> 1) * kArrQ + (wave & 1) * kBatch + pt] = f;
		}
	}
	if (left <= 64 && lane < kBatch - 64) // second slot skipped: its staging rows must still read as zero
		tile[(wave >> 1) * kArrQ + (wave & 1) * kBatch + 64 + lane] = uint4{0, 0, 0, 0};
	fin_commit(fz, fpre, fcache);
	__syncthreads();
	BN_TS(3);
	uint32_t acc[32];
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;
	if (wave == 0) {
		const eval_layout lay = make_eval_layout(lane);
		eval_staged(tile, lay, acc, []() {});
	}
	BN_TS(5);
	re9::tail<4>(acc, lane < 63, c, g, wave, lane, out, fz, seq, &fcache);
	BN_TS(8);
}

// ---------------------------------------------------------------------------------------------
// Resident tail.  Once the arrays are small a round costs ~17 us of kernel latency plus ~11 us of
// launch + completion round trip, for microseconds of arithmetic.  k_foldeval_tail runs the round
// it is launched for and then STAYS on the device: one workgroup of 8 waves, parked on a command
// word in fine-grained pinned host memory.  The host side of the next rounds (abi.cpp, "tail")
// checks that the next fold + evaluation request is the one this kernel will execute (same arrays,
// half the size, same finalize recipe), then just writes (z, command) and waits for the result
// mailbox -- a 1.8 us ping-pong instead of a launch (tools/launch_latency.hip).  Anything else
// cancels the kernel; so does a bounded spin (a host that went away cannot hang the GPU).
//
// All rounds after the first fold in place on fa.out.  Within one workgroup the folded values of
// round k are ordered before the loads of round k+1 by the workgroup barrier.
constexpr int kTailWaves = 8;
__global__ __launch_bounds__(64 * kTailWaves) void k_foldeval_tail(foldeval_args fa, uint64_t n_in, f128 z, f128 *out, fin_fuse fz,
                                                                  const uint64_t *cmd, uint64_t *status, uint64_t tail_id)
{
	extern __shared__ __attribute__((aligned(16))) uint4 tail_tile[]; // [kTailWaves][kWaveQ]
	__shared__ ctable_smem tab;
	__shared__ uint64_t next_z[2];
	__shared__ unsigned go;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned g = lane / 9, c = lane - g * 9;
	uint64_t round = 0;
	foldeval_args cur = fa;
	for (;;) {
		ctable_build(tab, z);
		uint32_t acc[32];
		foldeval_wave<false>(cur, n_in, tab, tail_tile + wave * kWaveQ, wave, kTailWaves, acc);
		re9::tail<kTailWaves>(acc, lane < 63, c, g, wave, lane, out, fz, fz.args.seq + round); // single workgroup: it is the "last" one
		if (n_in <= 4) break; // the fold to one element has no evaluation behind it
		// ---- park: wait for (tail_id, round + 1) or a cancel
		round++;
		if (threadIdx.x == 0) {
			unsigned ok = 0;
			for (uint32_t spins = 0; spins < (1u << 21); spins++) {
				const uint64_t w = __hip_atomic_load(cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
				if (w == ((tail_id << 20) | round)) {
					next_z[0] = __hip_atomic_load(cmd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					next_z[1] = __hip_atomic_load(cmd + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					ok = 1;
					break;
				}
				if (w == ((tail_id << 20) | 0xFFFFFu)) break; // cancelled
				__builtin_amdgcn_s_sleep(2);
			}
			go = ok;
		}
		__syncthreads();
		if (!__builtin_amdgcn_readfirstlane(go)) break; // (readfirstlane: keeps the loop state wave-uniform = in SGPRs)
		{
			const volatile uint32_t *zw = reinterpret_cast<const volatile uint32_t *>(next_z);
			const uint32_t z0 = __builtin_amdgcn_readfirstlane(zw[0]), z1 = __builtin_amdgcn_readfirstlane(zw[1]);
			const uint32_t z2 = __builtin_amdgcn_readfirstlane(zw[2]), z3 = __builtin_amdgcn_readfirstlane(zw[3]);
			z = f128{(uint64_t)z0 | ((uint64_t)z1 << 32), (uint64_t)z2 | ((uint64_t)z3 << 32)};
		}
		// next round: in place on the folded halves
		n_in >>= 1;
		cur.x0[0] = fa.out[0];
		cur.x0[1] = fa.out[1];
		cur.x1[0] = (const uint4 *)fa.out[0] + (n_in >> 1);
		cur.x1[1] = (const uint4 *)fa.out[1] + (n_in >> 1);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		__threadfence_system();
		__hip_atomic_store(status, tail_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// which of the two kernels launch_foldeval9 picks for this size (abi.cpp labels the launch with it)
bool foldeval9_is_small(int n_cu, uint64_t n_in)
{
	const uint64_t n = n_in >> 2;
	return (n + kBatch - 1) / kBatch <= (uint64_t)n_cu * 2;
}

// For both arrays j: out_j[i] = x0_j[i] + z * (x1_j[i] - x0_j[i]), i < n_in/2 (out_j may be x0_j), and
// accumulate the next round's (S_1, S_inf) of out_0 * out_1 into d_out[0], d_out[1].  n_in >= 4.
hipError_t launch_foldeval9(hipStream_t s, int n_cu, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse *fuse,
                            const arm_args *armed)
{
	arm_args arm{};
	if (armed) arm = *armed;
	if (n_in < 4 || (n_in & 3)) return hipErrorNotSupported;
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	const uint64_t n = n_in >> 2;
	const uint64_t n_batches = (n + kBatch - 1) / kBatch;
	uint64_t blocks = (n_batches + 3) / 4;
	const uint64_t cap = (uint64_t)n_cu * 2;
	if (foldeval9_is_small(n_cu, n_in)) {
		// small round: one workgroup per batch, the four waves share the fold (latency, not throughput)
		if (fa.scale_mask)
			hipLaunchKernelGGL(k_foldeval9_small<true>, dim3((unsigned)n_batches), dim3(256), 0, s, fa, n_in, z, d_out, fz, arm);
		else
			hipLaunchKernelGGL(k_foldeval9_small<false>, dim3((unsigned)n_batches), dim3(256), 0, s, fa, n_in, z, d_out, fz, arm);
		return hipGetLastError();
	}
	if (blocks > cap) blocks = cap;
	if (fa.scale_mask)
		hipLaunchKernelGGL((k_foldeval9<2, true>), dim3((unsigned)blocks), dim3(256), 0, s, fa, n_in, z, d_out, fz, arm);
	else
		hipLaunchKernelGGL((k_foldeval9<2, false>), dim3((unsigned)blocks), dim3(256), 0, s, fa, n_in, z, d_out, fz, arm);
	return hipGetLastError();
}

hipError_t launch_foldeval_tail(hipStream_t s, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse &fz,
                                const uint64_t *d_cmd, uint64_t *d_status, uint64_t tail_id)
{
	if (n_in < 8 || (n_in & 3) || !fz.counter || !fz.args.seq) return hipErrorNotSupported;
	const size_t lds = (size_t)kTailWaves * kWaveQ * sizeof(uint4);
	{
		const hipError_t e = func_lds_limit(reinterpret_cast<const void *>(&k_foldeval_tail), (int)lds);
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(k_foldeval_tail, dim3(1), dim3(64 * kTailWaves), lds, s, fa, n_in, z, d_out, fz, d_cmd, d_status, tail_id);
	return hipGetLastError();
}

} // namespace bn
