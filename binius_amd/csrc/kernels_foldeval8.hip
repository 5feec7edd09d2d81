// binius_amd/csrc/kernels_foldeval8.hip -- TWO sumcheck rounds per launch for the latency-shaped (small) rounds.
//
// A small round is a dependent chain: challenge -> fold -> products -> sums -> host (~10 us of kernel however few the
// elements are, tools/small_round_phases.hip).  The chain can serve two rounds.  Let Y be the arrays after the fold(s)
// of this launch, M elements each, in quarters Y0 | Y1 | Y2 | Y3 of Q = M / 4.  This round's evaluations
// (v3/bivariate_product.rs:303-408) are sums over the half cube,
//     y_1   = sum a2 b2 + sum a3 b3                      y_inf = sum (a0+a2)(b0+b2) + sum (a1+a3)(b1+b3),
// and after the NEXT challenge z the arrays are Y' = (Y0 + z (Y0+Y2) | Y1 + z (Y1+Y3)), whose evaluations are
// quadratics in z with coefficients that are sums of the same kind (character 2: P(1) = P0 + P1 + P2):
//     y_1'(z)   = P0 + z P1 + z^2 P2,   P0 = sum a1 b1,            P(1) = sum a3 b3,            P2 = sum (a1+a3)(b1+b3)
//     y_inf'(z) = Q0 + z Q1 + z^2 Q2,   Q0 = sum (a0+a1)(b0+b1),   Q(1) = sum (a2+a3)(b2+b3),   Q2 = sum (a0+..+a3)(b0+..+b3)
// Eight sums over Q points instead of two over 2 Q -- the same number of products as this round and the next together
// would cost (2 M against M + M/2 is a third more), but ONE chain: the host answers the next round itself (four scalar
// multiplications, abi_kernels.cpp) and the launch after that folds TWICE (both challenges are known by then):
//     X' = X0 + z1 (X0 + X1)  (n_in/2 elements),   Y = X'0 + z2 (X'0 + X'1)  (n_in/4 elements, written in place; the upper
//     half of X' is written too, so that memory ends up exactly as after two separate in-place folds).
//
// One workgroup of 8 waves per 64 quarter-points = 512 elements of Y, one per thread: (array, quarter) = wave, point =
// lane; 2^n_folds coalesced 16-byte loads per thread (issued before an armed launch waits for its challenges), the
// nibble tables of z1 and z2 built side by side by the two halves of the workgroup, the folded element to memory and
// into an LDS stage.  Waves 0..3 then take one PAIR of sums each with the [H*H | (L+H)*(L+H)] packing of the 9-lane
// kernels (re9.hpp), reading their rows from the stage:
//     wave 0: H = Y2, L = Y0 -> s0 = a2 b2,          s2 = (a0+a2)(b0+b2)        wave 2: H = Y1,    L = Y0    -> s4 = P0,   s5 = Q0
//     wave 1: H = Y3, L = Y1 -> s1 = a3 b3 = P(1),   s3 = (a1+a3)(b1+b3) = P2   wave 3: H = Y2+Y3, L = Y0+Y1 -> s6 = Q(1), s7 = Q2
// so the four products of a batch run on four SIMDs at once.  Accumulator slot 2 w + stream: the finalize recipe of this
// kernel is fixed (eight values = coefficient x slot, abi_kernels.cpp), the host combines.
#include <hip/hip_runtime.h>

#include "arm.hpp"
#include "ctable.hpp"
#include "re9.hpp"

namespace bn {

namespace {
constexpr int kPts = 64;                 // quarter-points per workgroup
constexpr int kG8 = 4;                   // 9-lane groups per evaluating wave (16 points each)
constexpr int kRowsPad = 65;             // stage rows per (array, quarter) block: 64 + 1 (blocks 16 banks apart)
constexpr int kBlk = re9::kBlkQ;         // 9 uint4 per (limb, group) block of the exchange tile
constexpr int kZero8 = 8 * kG8;          // zero block index
constexpr int kTile8 = (kZero8 + 1) * kBlk;

// tag contribution of one mirrored element (host tail; the host computes the same: abi_kernels.cpp): the two words rotated by
// amounts that depend on the element's index, so that neither a stale element nor two swapped ones leave the XOR unchanged
__device__ __forceinline__ uint64_t mirror_mix(uint64_t lo, uint64_t hi, uint64_t idx)
{
	const unsigned r1 = (unsigned)(idx & 63), r2 = (unsigned)((idx * 7 + 17) & 63);
	return ((lo << r1) | (r1 ? lo >> (64 - r1) : 0)) ^ ((hi << r2) | (r2 ? hi >> (64 - r2) : 0)) ^ (idx + 1) * 0x9E3779B97F4A7C15ull;
}

struct lay8 {
	unsigned off_a[4], off_b[4];
	unsigned off_w;
	bool loader, live;
	unsigned g, c;
};

__device__ __forceinline__ lay8 make_lay8(unsigned lane)
{
	lay8 l;
	l.g = lane / 9;
	l.c = lane - l.g * 9;
	l.live = lane < 9 * kG8;
	l.loader = l.live && l.c < 8;
	const unsigned mask = l.live ? re9::combo_mask(l.c) : 0u;
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const bool use = (mask >> s) & 1;
		l.off_a[s] = (use ? (unsigned)(s * kG8 + l.g) : (unsigned)kZero8) * kBlk;
		l.off_b[s] = (use ? (unsigned)((4 + s) * kG8 + l.g) : (unsigned)kZero8) * kBlk;
	}
	l.off_w = (l.loader ? (l.c * kG8 + l.g) : 0u) * kBlk;
	return l;
}

// One pair of sums for one wave: rows H (and H2), L (and L2) of the stage, n_valid points; leaves the lane's 32
// accumulator planes in acc (low 16 bits: H*H, high 16 bits: (L+H)*(L+H)).
__device__ __forceinline__ void eval_pair(const uint4 *stage, uint4 *wt, const lay8 &lay, unsigned h0, int h1, unsigned l0, int l1, unsigned n_valid,
                                          uint32_t (&acc)[32])
{
	const uint32_t *stw = reinterpret_cast<const uint32_t *>(stage);
	const unsigned lane = threadIdx.x & 63;
	if (lane < kBlk) wt[kZero8 * kBlk + lane] = uint4{0, 0, 0, 0};
	// loader lane (g, c): word column c & 3 of array c >> 2, rows 4 j + g (banks: 4 g + (c & 3) + 16 (c >> 2): all distinct)
	const unsigned a = (lay.c >> 2) & 1, wc = lay.c & 3, g = lay.live ? lay.g : 0;
	const unsigned bh0 = ((a * 4 + h0) * kRowsPad + g) * 4 + wc, bl0 = ((a * 4 + l0) * kRowsPad + g) * 4 + wc;
	const unsigned bh1 = ((a * 4 + (unsigned)(h1 < 0 ? 0 : h1)) * kRowsPad + g) * 4 + wc, bl1 = ((a * 4 + (unsigned)(l1 < 0 ? 0 : l1)) * kRowsPad + g) * 4 + wc;
	uint32_t r[32];
#pragma unroll
	for (int j = 0; j < 16; j++) {
		uint32_t h = stw[bh0 + 16 * j], l = stw[bl0 + 16 * j];
		if (h1 >= 0) { // (wave-uniform)
			h ^= stw[bh1 + 16 * j];
			l ^= stw[bl1 + 16 * j];
		}
		const bool ok = (unsigned)(4 * j) + g < n_valid;
		r[j] = ok ? h : 0u;
		r[16 + j] = ok ? (l ^ h) : 0u;
	}
	BN_TS(4);
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
	}
	BN_TS(5);
	bs_mul<5>(A, B, acc);
	BN_TS(6);
}

// Workgroup tail for eight sums: wave w < 4 holds the planes of slots 2 w (low halves) and 2 w + 1 (high halves).
// Collapse, recombine the nine limb products per slot, XOR into out[0..8) (or keep them in LDS when the launch is a
// single workgroup) and run the fused finalize in the last workgroup.  All 512 threads.
__device__ __forceinline__ void tail8(const uint32_t (&acc)[32], const lay8 &lay, unsigned wave, unsigned lane, f128 *out, const fin_fuse &fz, uint64_t seq,
                                      const fin_cache &fc, uint64_t *tag_acc = nullptr)
{
	__shared__ uint32_t red[4][2][9][kG8];
	__shared__ uint64_t wsum[4][4];
	if (wave < 4) {
		uint32_t s_lo = 0, s_hi = 0;
#pragma unroll
		for (int p = 0; p < 32; p++) {
			s_lo |= (__popc(acc[p] & 0xFFFFu) & 1u) << p;
			s_hi |= (__popc(acc[p] >> 16) & 1u) << p;
		}
		if (lay.live) {
			red[wave][0][lay.c][lay.g] = s_lo;
			red[wave][1][lay.c][lay.g] = s_hi;
		}
	}
	__syncthreads();
	if (wave < 4 && lane < 2) {
		uint32_t pc[9];
#pragma unroll
		for (int cc = 0; cc < 9; cc++) {
			uint32_t v = 0;
#pragma unroll
			for (int gg = 0; gg < kG8; gg++)
				v ^= red[wave][lane][cc][gg];
			pc[cc] = v;
		}
		const uint64_t Z0 = re9::combine32(pc[0], pc[1], pc[2]);
		const uint64_t Z2 = re9::combine32(pc[3], pc[4], pc[5]);
		const uint64_t Z1p = re9::combine32(pc[6], pc[7], pc[8]);
		const f128 S = re9::combine64(Z0, Z2, Z1p);
		wsum[wave][2 * lane] = S.lo;
		wsum[wave][2 * lane + 1] = S.hi;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (a host-tail launch: every wave's stores into the host staging have left before the barrier)
	__syncthreads();
	BN_TS(7);
	unsigned *const counter = fc.counter;
	if (counter && gridDim.x == 1 && out == fc.S) {
		// single workgroup: the sums never leave the chip (wsum is laid out exactly as f128 S_local[8])
		finalize_cached(fc, seq, reinterpret_cast<const f128 *>(&wsum[0][0]));
		return;
	}
	if (threadIdx.x < 16) {
		const uint64_t v = wsum[threadIdx.x >> 2][threadIdx.x & 3];
		if (v) atomicXor(reinterpret_cast<unsigned long long *>(out) + threadIdx.x, (unsigned long long)v);
	}
	if (counter) {
		// same ticket protocol as re9::tail (device-scope atomics only, no fences)
		__shared__ unsigned is_last;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (threadIdx.x == 0) {
			const unsigned t = atomicAdd(counter, 1u);
			is_last = (t == gridDim.x - 1) ? 1u : 0u;
		}
		__syncthreads();
		if (is_last) {
			if (tag_acc && threadIdx.x == 0) {
				// a host-tail launch of several workgroups: every one of them XORed the tag of its part of the staging into
				// tag_acc before it took its ticket; published here, by the lane that publishes the sequence word afterwards
				const uint64_t t = __hip_atomic_load(tag_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(&fc.mail[66].lo, t ^ seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(tag_acc, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			finalize_cached(fc, seq);
			if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}
} // namespace

template <int NF>
__global__ __launch_bounds__(512, 1) void k_foldeval8(foldeval8_args fa, f128 z1, f128 z2, f128 *out, fin_fuse fz, arm_args arm)
{
	__shared__ uint4 stage[8 * kRowsPad];
	__shared__ uint4 tile[4][kTile8];
	__shared__ ctable_smem tab[NF == 2 ? 2 : 1];
	__shared__ uint4 phi_T[512]; // host tail: the nibble table of the host's basis change (ctable.hpp layout of T)
	__shared__ uint64_t mir_tag[8];
	__shared__ fin_cache fcache;
	BN_TS(0);
	const uint64_t seq = fz.args.seq;
	const unsigned tid = threadIdx.x, lane = tid & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint64_t m = fa.n_in >> NF, q = m >> 2; // elements of Y, quarter-points
	const uint64_t p0 = (uint64_t)blockIdx.x * kPts;
	const unsigned n_valid = (unsigned)((q - p0) < (uint64_t)kPts ? (q - p0) : (uint64_t)kPts);
	// this thread's element of Y: array wave >> 2, quarter wave & 3, point lane
	const unsigned arr = wave >> 2, qt = wave & 3;
	const uint64_t i = (uint64_t)qt * q + p0 + lane;
	const bool have = lane < n_valid;
	uint4 v[4] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}};
	if (have) {
		const uint4 *x0 = (const uint4 *)fa.x0[arr], *x1 = (const uint4 *)fa.x1[arr];
		if constexpr (NF == 0) {
			v[0] = i < (m >> 1) ? x0[i] : x1[i - (m >> 1)]; // Y itself: x0 | x1 are its halves
		} else if constexpr (NF == 1) {
			v[0] = x0[i];
			v[1] = x1[i];
		} else {
			v[0] = x0[i];
			v[1] = x1[i];
			v[2] = x0[i + m];
			v[3] = x1[i + m];
		}
	}
	const fin_pref fpre = fin_prefetch(fz);
	const bool to_host = fa.mirror != nullptr; // (uniform) host tail: Y also goes to the host, in the host's basis
	uint4 phi_v{0, 0, 0, 0};
	if (to_host) phi_v = fa.phi_tab[tid];
	BN_TS(1);
	if (arm.h_cmd) { // (uniform) armed launch: the data is on its way, the challenges are what is missing (arm.hpp)
		f128 z2_in;
		if (!arm_wait(arm, z1, z2_in)) return;
		z2 = z2_in;
	}
	if (to_host) {
		phi_T[tid] = phi_v;
		if constexpr (NF == 0) __syncthreads(); // (the table builds below contain the barrier otherwise)
	}
	if constexpr (NF == 2)
		ctable_build_group(tab[tid >> 8], (tid >> 8) ? z2 : z1, tid & 255, 256); // both tables at once
	else if constexpr (NF == 1)
		ctable_build(tab[0], z1);
	BN_TS(2);
	uint4 y = v[0], u_hi{0, 0, 0, 0};
	uint64_t tag = 0;
	if constexpr (NF >= 1) y = xor4(v[0], ctable_mul(tab[0], xor4(v[0], v[1])));
	if constexpr (NF == 2) {
		u_hi = xor4(v[2], ctable_mul(tab[0], xor4(v[2], v[3]))); // X'[i + m]
		y = xor4(y, ctable_mul(tab[1], xor4(y, u_hi)));
	}
	if (have) {
		if constexpr (NF >= 1) ((uint4 *)fa.out[arr])[i] = y;
		if constexpr (NF == 2) ((uint4 *)fa.out[arr])[i + m] = u_hi; // memory ends up exactly as after two separate in-place folds
		stage[(arr * 4 + qt) * kRowsPad + lane] = y;
		if (to_host) {
			// Phi is GF(2)-linear: one more nibble-table product.  PLAIN 16-byte stores into the pinned staging -- posted writes,
			// a wave's 64 of them one coalesced burst (system-scope atomic stores of the 1024 words were measured at ~120 ns
			// each, one PCIe round trip after the other: 125 us per hand-over) --, and nothing is assumed about the order in
			// which they and the sequence word reach host memory: the staging validates itself (mirror_tag below).
			const uint4 py = ctable_mul(*reinterpret_cast<const ctable_smem *>(phi_T), y);
			reinterpret_cast<uint4 *>(fa.mirror)[(uint64_t)arr * m + i] = py;
			tag = mirror_mix((uint64_t)py.x | ((uint64_t)py.y << 32), (uint64_t)py.z | ((uint64_t)py.w << 32), (uint64_t)arr * m + i);
		} else if (m == 4 && fz.mail) {
			// the last launch of a sumcheck: Y is four elements per array -- the host folds them itself (six products) when the
			// caller reads the final evaluations; published before the sequence number (drained below, a barrier follows)
			f128 *slot = fz.mail + 32 + arr * 4 + qt;
			__hip_atomic_store(&slot->lo, (uint64_t)y.x | ((uint64_t)y.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(&slot->hi, (uint64_t)y.z | ((uint64_t)y.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
	}
	if (to_host) {
		// the staging's tag: XOR over the elements of a 64-bit mix of (index, value), published in mailbox word 66 by the lane that
		// publishes the sequence word later (same lane: the tag is out first); the host accepts the staging only when the tag it
		// reads is the tag of the data it reads (abi_kernels.cpp)
#pragma unroll
		for (int sh = 32; sh >= 1; sh >>= 1) tag ^= __shfl_xor(tag, sh, 64);
		if (lane == 0) mir_tag[wave] = tag;
	}
	fin_commit(fz, fpre, fcache);
	__syncthreads();
	if (to_host && tid == 0) {
		uint64_t t = 0;
#pragma unroll
		for (int w = 0; w < 8; w++) t ^= mir_tag[w];
		if (gridDim.x == 1)
			__hip_atomic_store(&fz.mail[66].lo, t ^ seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		else if (t)
			atomicXor(reinterpret_cast<unsigned long long *>(fa.tag_acc), (unsigned long long)t); // (the last workgroup publishes: tail8)
	}
	BN_TS(3);
	uint32_t acc[32];
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;
	const lay8 lay = make_lay8(lane);
	if (wave < 4) {
		// (wave-uniform selectors)       H        H2                L        L2
		const unsigned h0 = wave == 0 ? 2u : (wave == 1 ? 3u : (wave == 2 ? 1u : 2u));
		const unsigned l0 = wave == 1 ? 1u : 0u;
		eval_pair(stage, tile[wave], lay, h0, wave == 3 ? 3 : -1, l0, wave == 3 ? 1 : -1, n_valid, acc);
	}
	tail8(acc, lay, wave, lane, out, fz, seq, fcache, to_host && gridDim.x > 1 ? fa.tag_acc : nullptr);
	BN_TS(8);
}

// Y = the arrays after fa.n_folds folds (0: x0 | x1 are the halves of Y, nothing is written; 1: X0 + z1 (X0 + X1);
// 2: that folded once more with z2), M = n_in >> n_folds elements each, M >= 4 a multiple of 4; d_out[0..8) ^= the
// eight sums in slot order (header comment).  The stage holds one 64-point batch per workgroup, one workgroup per CU:
// the launcher's own bound is far above what the dispatcher sends (abi_kernels.cpp two_round_size_ok).
hipError_t launch_foldeval8(hipStream_t s, const foldeval8_args &fa, f128 z1, f128 z2, f128 *d_out, const fin_fuse *fuse, const arm_args *armed)
{
	if (fa.n_folds > 2) return hipErrorNotSupported;
	const uint64_t m = fa.n_in >> fa.n_folds;
	if (m < 4 || (m & 3) || (m << fa.n_folds) != fa.n_in) return hipErrorNotSupported;
	const uint64_t q = m >> 2, blocks = (q + kPts - 1) / kPts;
	if (blocks > 4096) return hipErrorNotSupported;
	if (fa.mirror && (!fa.phi_tab || (blocks > 1 && !fa.tag_acc))) return hipErrorNotSupported; // (several workgroups accumulate the staging's tag on the device)
	arm_args arm{};
	if (armed) arm = *armed;
	fin_fuse fz{};
	if (fuse) fz = *fuse;
	if (!fz.counter) return hipErrorNotSupported; // the eight sums are always finalized in the kernel
	switch (fa.n_folds) {
	case 0: hipLaunchKernelGGL(k_foldeval8<0>, dim3((unsigned)blocks), dim3(512), 0, s, fa, z1, z2, d_out, fz, arm); break;
	case 1: hipLaunchKernelGGL(k_foldeval8<1>, dim3((unsigned)blocks), dim3(512), 0, s, fa, z1, z2, d_out, fz, arm); break;
	default: hipLaunchKernelGGL(k_foldeval8<2>, dim3((unsigned)blocks), dim3(512), 0, s, fa, z1, z2, d_out, fz, arm); break;
	}
	return hipGetLastError();
}

} // namespace bn
