// binius_amd/csrc/kernels_group.hip -- ONE launch for a whole batch round of bivariate sumchecks: every product claim of every
// prover that is ready becomes a JOB of the launch -- the fold of the claim's two multilinears fused with the evaluation of the
// next round polynomial (kind 0), or the evaluation alone (kind 1) -- and the launch returns the RAW sums of all jobs.
//
//   kind 0   a'[i] = a[i] + z (a[i + N/2] - a[i]), likewise b'          (extrapolate_line, compute/src/layer.rs:421)
//            S_1   = sum_{j < N/4} a'[j + N/4] b'[j + N/4]
//            S_inf = sum_{j < N/4} (a'[j] + a'[j + N/4]) (b'[j] + b'[j + N/4])
//   kind 1   S_1 = sum_j a_1[j] b_1[j], S_inf = sum_j (a_0[j] + a_1[j]) (b_0[j] + b_1[j]) over the halves a_0 | a_1, b_0 | b_1 as they are
//   kind 2   two plain inner products of rows: S[slot] = sum_j A[j] B[j], S[slot + 1] = sum_j C[j] D[j]  (the final sums of compiled
//            circuits for all evaluation points of an old-HAL request in one launch, abi_hal.cpp; C = null: one product;
//            B or D = null: the all-ones row, i.e. the plain sum of A or C)
//   kind 3   the fold of kind 0 for one or two arrays and nothing else (an array that several claims share is folded ONCE, by a
//            kind-0 job of one of its claims or by a kind-3 job; the other claims over it are kind-1 / kind-4 jobs)
//   kind 4   kind 0 for the FIRST array only: a is folded on the way, b is read as it is (its halves b_0 | b_1, like kind 1) -- a claim
//            whose second multilinear is shared with other claims (piop::prove pairs every committed multilinear of a size with
//            the few ring-switch transparents of that size: a star around each transparent) folds its own array and reads the
//            shared one, already folded by an earlier job of the chain or up to date
//
// Chains.  A kind-1 job may read what a kind-0 / kind-3 job of the SAME launch writes -- the folds are in place, so a workgroup
// that evaluated such a claim while another one folded the array would read half-folded memory.  Jobs that depend on each other
// therefore form a chain: ONE set of workgroups runs them one after the other, each workgroup on the same tiles in every job,
// the folding jobs first.  A workgroup then only ever reads back what it wrote itself: it waits for its stores (vmcnt), meets
// its own waves at a barrier and invalidates its vector cache (the `acquire` flag of the first reading job) -- no grid-wide
// synchronisation, no second launch, and the shared arrays' folds cost no pass of their own.
//
// Why.  The reference's PCS prover issues k product claims over m multilinears per prover and runs several provers front-loaded
// on one ComputeLayer: per batch round  execute(P_1) .. execute(P_p), one challenge, fold(P_1) .. fold(P_p)
// (core/src/piop/prove.rs:271-287, core/src/protocols/sumcheck/prove/front_loaded.rs:122-155,
// v3/bivariate_product.rs:303-408).  kernels_foldeval_fp4.hip fuses exactly ONE claim over two arrays.  Here the workgroups of
// one launch are dealt out to the jobs (job j owns workgroups [wg_begin, wg_begin + wg_count): its own tile order, its own
// nibble table for its own challenge, its own pair of accumulator slots); inside its range a workgroup is the twelve-wave
// workgroup of kernels_foldeval_fp4.hip -- waves 4 .. 11 fold / load and stage a pair of tiles, waves 0 .. 3 run the FP4 Gram
// k-steps of the previous pair (gram_fp4.hpp).  The last workgroup of the LAUNCH (device-scope ticket) copies all accumulator
// slots to the pinned mailbox and re-zeroes them; the batch coefficients (value = init + sum_c alpha^c S_c,
// cpu/layer.rs:512) are applied by the host, which keeps the kernel independent of the caller's recipe -- so the sums of a
// prover whose execute() has not been called yet can be computed in the same launch (abi_group.cpp).
//
// Algorithmic bytes per job: kind 0 read 16 * 2 * N + write 8 * 2 * N = 48 N (24 * m * N summed over a prover's disjoint claims);
// kind 1 read 16 * 2 * (N / 2) = 16 N; kind 3 24 N per array; kind 4 24 N + 8 N.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <queue>
#include <vector>

#include "ctable.hpp"
#include "gram_fp4.hpp"

namespace bn {

using namespace gram4;

namespace {

typedef unsigned int gq_v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 gq_load(const uint4 *p)
{
	if constexpr (NT) {
		const gq_v4u v = __builtin_nontemporal_load(reinterpret_cast<const gq_v4u *>(p));
		return uint4{v.x, v.y, v.z, v.w};
	} else {
		return *p;
	}
}
template <bool NT>
__device__ __forceinline__ void gq_store(uint4 *p, uint4 r)
{
	if constexpr (NT) {
		const gq_v4u v = {r.x, r.y, r.z, r.w};
		__builtin_nontemporal_store(v, reinterpret_cast<gq_v4u *>(p));
	} else {
		*p = r;
	}
}

constexpr unsigned kGramWaves = 4;  // waves 0 .. 3: the Gram k-steps (one per SIMD)
constexpr unsigned kFoldGroups = 2; // waves 4 .. 11: two groups of four waves, a tile per group
constexpr unsigned kThreads = 64 * kGramWaves * (1 + kFoldGroups);

// The kernel arguments.  The job table itself is NOT among them (256 jobs of 96 bytes: six times what a kernel-argument block
// holds): it lies in pinned, device-mapped memory (internal.hpp group_tables) that the host fills right before the launch, and
// a workgroup reads the one job it works on -- and the followers of its chain -- from there with scalar loads.  Which job that
// is comes from head_of_wg: two 16-bit job numbers per word, indexed by workgroup.
struct group_kargs {
	uint32_t head_of_wg[kGroupMaxGrid / 2];
	const group_job *table;
	f128 *S;           // accumulator slots (zero before the launch, zero after it)
	f128 *vals;        // pinned: [0, n_slots) the raw sums
	f128 *mail;        // pinned mailbox: word 64 the sequence number
	unsigned *counter; // device-scope ticket (zero between launches)
	uint64_t seq;
	uint32_t n_jobs, n_slots, prio, pad;
};

// read-only views for SCALAR loads with a run-time index: the kernel-argument segment and the pinned job table are both memory
// nobody writes while the launch runs (address space 4 = constant: a load through it with a uniform address is an s_load, its
// result lives in scalar registers -- through a generic pointer the compiler would issue vector loads and carry every pointer
// of the job as a 64-bit vector value through the tile loops)
typedef const group_job __attribute__((address_space(4))) *cjob_ptr;
typedef const uint32_t __attribute__((address_space(4))) *cu32_ptr;
__device__ __forceinline__ cu32_ptr kernarg_words()
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (cu32_ptr)__builtin_amdgcn_kernarg_segment_ptr();
#else
	return nullptr;
#endif
}
__device__ __forceinline__ cjob_ptr as_const_jobs(const group_job *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (cjob_ptr)p;
#else
	(void)p;
	return nullptr;
#endif
}

} // namespace

// what a workgroup needs to know about its job and its share of the job's tiles (uniform)
struct group_wg {
	const char *X0[2], *X1[2];
	char *OUT[2];
	f128 z;
	uint64_t n;
	uint32_t tbase, tstride, tlimit, t0;
	uint32_t adj; // 1: the two fold groups take ADJACENT tiles (2 p, 2 p + 1) of the workgroup's p-th pair instead of tiles a stride apart
};

// The loops of one workgroup for a job of kind KIND (0: fold + evaluate, 1: evaluate).  A function template per kind -- and not
// a run-time `kind` inside one loop -- because the loaded elements must live in registers: with both kinds' uses of x0 / x1 in
// one body the compiler keeps the two arrays in scratch memory and every tile pays 128 bytes of scratch stores and loads per
// lane (measured: the launch at half the rate of kernels_foldeval_fp4.hip).  Leaves the parity words of the Gram waves in Gc.
template <int KIND, bool FULL, bool NT>
__device__ __forceinline__ void group_loops(const group_wg &w, uint32_t *T_dyn, ctable_smem &tab, gram_parity &Gc, uint32_t prio, bool build)
{
	constexpr bool kFolds = KIND == 0 || KIND == 3 || KIND == 4; // the job folds its arrays (KIND 3: and nothing else -- no staging, no Gram work, no sums;
	                                                                // KIND 4: its first array only, the second is read as it is)
	constexpr bool kTwoAhead = KIND == 1 || KIND == 2;
	// (the thread index behind an opaque move: everything derived from it -- roles, lane offsets -- is recomputed per job of a chain
	// instead of being hoisted out of the kernel's job loop and kept in registers across all four loop bodies)
	unsigned tx = threadIdx.x;
	asm volatile("" : "+v"(tx));
	const unsigned lane = tx & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tx >> 6);
	const bool folds = wave >= kGramWaves;
	const unsigned grp = folds ? (wave - kGramWaves) >> 2 : 0;   // fold group: which tile of the pair
	const unsigned ftid = (tx - 64 * kGramWaves) & 255; // the lane's point inside its tile (fold waves)
	const uint64_t n = w.n;
	const uint32_t tbase = w.tbase, tstride = w.tstride, tlimit = w.tlimit, t0 = w.t0;
	// iteration variable `it` = t0, t0 + step, ... < limit; tile of fold group g in iteration it
	const uint32_t adj = w.adj, step = adj ? tstride : 2 * tstride, limit = adj ? (tlimit + 1) >> 1 : tlimit;
	auto tile_of = [&](uint32_t it, uint32_t g) { return adj ? 2 * it + g : it + g * tstride; };

	// quadrant q = 2 * side + half.  KIND 0: x0[q] / x1[q] = the two elements the fold of (side, half) reads; KIND 1: x0[q] = the
	// element of (side, half) itself.
	uint4 x0[4], x1[kFolds ? 4 : 1];
	const uint32_t voff = ftid * 16u;
	auto lane_off = [&]() { // (kernels_foldeval_fp4.hip: keeps the lane offset out of a loop-invariant 64-bit vector base)
		uint32_t v = voff;
		asm volatile("" : "+v"(v));
		return v;
	};
	uint32_t vo = lane_off();
	auto in_range = [&](uint32_t t) { return FULL || (uint64_t)(tbase + t) * kTP + ftid < n; };
	// KIND 1 keeps TWO tiles of loads in flight per fold group (x0: the tile after this one, xb: the one after that): with nothing to
	// compute between a tile's loads and its staging, one tile in flight is 32 KiB per CU and the launch sits at the memory
	// latency (2.1 TB/s measured at 2^25 points per claim); KIND 0 has 64 KiB of loads per CU in flight with one tile per group
	uint4 xb[kTwoAhead ? 4 : 1];
	auto load_into = [&](uint4 *dst0, uint32_t t, int q) {
		const uint32_t o = in_range(t) ? vo : 0u; // (a lane past the end reads the tile's first element: in range, never used)
		if constexpr (kFolds) {
			if (KIND == 3 && q >= 2 && w.X0[1] == nullptr) return; // (uniform) a fold-only job of ONE array
			if (KIND == 4 && q >= 2) { // the second array as it is: X0 = its lower half (evaluations at 0), X1 = its upper half
				const uint64_t e1 = (uint64_t)(tbase + t) * kTP * 16;
				dst0[q] = gq_load<NT>(reinterpret_cast<const uint4 *>((q & 1 ? w.X1[1] : w.X0[1]) + e1 + o));
				return;
			}
			const uint64_t e = ((q & 1 ? n : 0) + (uint64_t)(tbase + t) * kTP) * 16; // (uniform)
			dst0[q] = gq_load<NT>(reinterpret_cast<const uint4 *>(w.X0[q >> 1] + e + o));
			x1[q] = gq_load<NT>(reinterpret_cast<const uint4 *>(w.X1[q >> 1] + e + o));
		} else if constexpr (KIND == 1) {
			const uint64_t e1 = (uint64_t)(tbase + t) * kTP * 16;
			dst0[q] = gq_load<NT>(reinterpret_cast<const uint4 *>((q & 1 ? w.X1[q >> 1] : w.X0[q >> 1]) + e1 + o));
		} else {
			// rows A, B, C, D = x0[0], x0[1], x1[0], x1[1]; an absent second product reads the first one's rows (its sum is never looked at)
			const uint64_t e1 = (uint64_t)(tbase + t) * kTP * 16;
			const bool second = q >= 2 && w.X1[0] != nullptr; // (uniform)
			const char *row = q < 2 ? w.X0[q] : (second ? w.X1[q - 2] : w.X0[q - 2]);
			if (row) // (uniform) a null second factor is the all-ones row: the sum of the first factor, with nothing loaded for it
				dst0[q] = gq_load<NT>(reinterpret_cast<const uint4 *>(row + e1 + o));
			else
				dst0[q] = uint4{1, 0, 0, 0};
		}
	};
	auto load1 = [&](uint32_t t, int q) { load_into(x0, t, q); };
	const uint32_t tm0 = tile_of(t0, grp); // this fold group's first tile
	if (folds && t0 < limit && tm0 < tlimit) {
#pragma unroll
		for (int q = 0; q < 4; q++)
			load1(tm0, q);
		if constexpr (kTwoAhead) {
			const uint32_t tm1 = (t0 + step < limit && tile_of(t0 + step, grp) < tlimit) ? tile_of(t0 + step, grp) : tm0;
#pragma unroll
			for (int q = 0; q < 4; q++)
				load_into(xb, tm1, q);
		}
	}
	if constexpr (kFolds) {
		if (build) ctable_build(tab, w.z); // (uniform; the loads above are in flight meanwhile; ends with a barrier)
	}
	if (folds) {
		switch (prio & 3) { // the fold waves issue ahead of the Gram wave of their SIMD (kernels_foldeval_fp4.hip)
		case 1: __builtin_amdgcn_s_setprio(1); break;
		case 2: __builtin_amdgcn_s_setprio(2); break;
		case 3: __builtin_amdgcn_s_setprio(3); break;
		default: break;
		}
		const stage4_role sr = make_stage4_role(ftid);
		uint32_t *Tn = T_dyn + grp * kTile4W;
		unsigned buf = 0;
		// one pair of tiles: `cur` holds this group's tile, loaded one (KIND 0) or two (KIND 1) iterations ago, and is refilled for
		// the iteration that will use it next
		auto pair_step = [&](uint32_t t, uint4 *cur) {
			const uint32_t tm = tile_of(t, grp);
			if (tm < tlimit) { // (uniform; false only for group 1 on an odd last pair)
				// the last iteration(s) re-request their own tile (cache hits) instead of branching around the loads
				constexpr uint32_t kAhead = kTwoAhead ? 2 : 1; // iterations ahead
				const uint32_t ta = t + kAhead * step;
				const uint32_t tn = (ta < limit && tile_of(ta, grp) < tlimit) ? tile_of(ta, grp) : tm;
				vo = lane_off();
				const bool ok = in_range(tm);
				uint4 f[4];
				if constexpr (kFolds) {
					const uint64_t pt16 = (uint64_t)(tbase + tm) * kTP * 16; // (uniform)
					const int nq = (KIND == 3 && w.X0[1] == nullptr) ? 2 : 4; // (uniform)
#pragma unroll
					for (int q = 0; q < 4; q++) {
						if (q >= nq) break;
						if (KIND == 4 && q >= 2)
							f[q] = x0[q];
						else
							f[q] = ctable_mul_acc<8, true>(tab, xor4(x0[q], x1[q]), x0[q]);
						load1(tn, q); // this quadrant of the group's next tile flies from here on
					}
					if (FULL || ok) {
#pragma unroll
						for (int q = 0; q < 4; q++) {
							if (q >= nq || (KIND == 4 && q >= 2)) break;
							gq_store<NT>(reinterpret_cast<uint4 *>(w.OUT[q >> 1] + (q & 1 ? n * 16 : 0) + pt16 + vo), f[q]);
						}
					}
				} else {
#pragma unroll
					for (int q = 0; q < 4; q++) {
						f[q] = cur[q];
						load_into(cur, tn, q);
					}
				}
				if constexpr (KIND != 3) {
					if (!FULL && !ok) {
#pragma unroll
						for (int q = 0; q < 4; q++)
							f[q] = uint4{0, 0, 0, 0}; // points past the end carry zeros
					}
				}
				if constexpr (KIND == 3) {
					// (nothing is staged)
				} else if constexpr (KIND == 2) {
					// sets 0 / 1 = the first product's rows, sets 2 / 3 = the second's
					stage4_elem(Tn, sr, 0, f[0]);
					stage4_elem(Tn, sr, 1, f[1]);
					stage4_elem(Tn, sr, 2, f[2]);
					stage4_elem(Tn, sr, 3, f[3]);
				} else {
					// half 1 is the evaluation at 1, half 0 its partner: sets 0 / 1 = u, v at 1; sets 2 / 3 = u, v at infinity
					stage4_elem(Tn, sr, 0, f[1]);
					stage4_elem(Tn, sr, 2, xor4(f[1], f[0]));
					stage4_elem(Tn, sr, 1, f[3]);
					stage4_elem(Tn, sr, 3, xor4(f[3], f[2]));
				}
			}
			if constexpr (KIND != 3) __syncthreads(); // the pair is staged; the Gram waves are done with the buffer this wave writes next
			buf ^= 1;
			Tn = T_dyn + (buf * kFoldGroups + grp) * kTile4W;
		};
		if constexpr (kFolds) {
			for (uint32_t t = t0; t < limit; t += step) pair_step(t, x0);
		} else { // (KIND 1, 2: two tiles in flight)
			for (uint32_t t = t0; t < limit;) {
				pair_step(t, x0);
				t += step;
				if (t >= limit) break;
				pair_step(t, xb);
				t += step;
			}
		}
		__builtin_amdgcn_s_setprio(0);
	} else if constexpr (KIND != 3) {
		const gram4_role gr = make_gram4_role(wave, lane);
		v16f acc[kAccTiles];
		acc4_zero(acc);
		unsigned buf = 0;
		for (uint32_t t = t0; t < limit; t += step) {
			__syncthreads();
			const uint32_t *Tp = T_dyn + buf * kFoldGroups * kTile4W;
			gram4_tile(Tp, gr, acc);
			if (tile_of(t, 1) < tlimit) gram4_tile(Tp + kTile4W, gr, acc);
			buf ^= 1;
		}
		parity4(acc, gr, wave, lane, Gc);
	}
}

template <bool FULL, bool NT>
__global__ __launch_bounds__(kThreads, 1) void k_group_fp4(group_kargs ga)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t T_dyn[]; // 2 buffers x 2 tiles of FP4 operands
	__shared__ ctable_smem tab;
	__shared__ gram_parity Gc;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

	// ---- this workgroup's job (uniform): the head of its chain, or the lone job, from the table in the kernel arguments; the job
	// itself (and the followers of the chain) from the pinned table
	static_assert(offsetof(group_kargs, head_of_wg) == 0, "the head-of-workgroup table leads the kernel-argument block");
	const uint32_t hw = kernarg_words()[blockIdx.x >> 1];
	const unsigned ji = (blockIdx.x & 1) ? hw >> 16 : hw & 0xFFFFu;
	const cjob_ptr head = as_const_jobs(ga.table) + ji;
	const uint32_t G = head->wg_count, b = blockIdx.x - head->wg_begin;
	const bool xcd_order = (G & 7) == 0 && (head->wg_begin & 7) == 0;
	const unsigned n_sub = 1 + head->chain;
	__shared__ uint64_t z3[2][3];
	__shared__ f128 s_loc[2];
	__shared__ unsigned is_last;
	const unsigned tid = threadIdx.x;
	f128 z_built{0, 0};
	bool have_table = false;
	for (unsigned sub = 0; sub < n_sub; sub++) {
		const cjob_ptr jb = head + sub;
		if (jb->acquire) {
			// this workgroup's own stores of the chain's earlier jobs (folded arrays, in place) are read back by this job: all of
			// them have reached the L2 (vmcnt), and the vector cache forgets the lines it loaded before they were written
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		}
		group_wg w;
		w.n = jb->n; // evaluation points of the job
		w.z = f128{jb->z.lo, jb->z.hi};
		for (int sd = 0; sd < 2; sd++) {
			w.X0[sd] = reinterpret_cast<const char *>(jb->x0[sd]);
			w.X1[sd] = reinterpret_cast<const char *>(jb->x1[sd]);
			w.OUT[sd] = reinterpret_cast<char *>(jb->out[sd]);
		}
		const uint32_t n_tiles = (uint32_t)(FULL ? w.n / kTP : (w.n + kTP - 1) / kTP);
		// tile order inside the job's range: as kernels_foldeval_mfma.hip (XCD x = blockIdx.x & 7 takes the x-th contiguous eighth of
		// the job's tiles) when the range starts on a multiple of eight workgroups and is a multiple of eight long
		w.tbase = 0;
		w.tstride = G;
		w.tlimit = n_tiles;
		w.t0 = b;
		if (xcd_order) {
			const uint32_t chunk = (n_tiles + 7) >> 3;
			w.tbase = (b & 7) * chunk;
			w.tstride = G >> 3;
			w.t0 = b >> 3;
			w.tlimit = w.tbase >= n_tiles ? 0 : (n_tiles - w.tbase < chunk ? n_tiles - w.tbase : chunk);
		}
		w.adj = (ga.prio >> 2) & 1;
		const uint32_t kind = jb->kind; // (uniform)
		const bool build = !have_table || !(z_built == w.z);
		if (kind == 0 || kind == 3 || kind == 4) {
			z_built = w.z;
			have_table = true;
		}
		if (kind == 0)
			group_loops<0, FULL, NT>(w, T_dyn, tab, Gc, ga.prio, build);
		else if (kind == 4)
			group_loops<4, FULL, NT>(w, T_dyn, tab, Gc, ga.prio, build);
		else if (kind == 1)
			group_loops<1, FULL, NT>(w, T_dyn, tab, Gc, ga.prio, false);
		else if (kind == 2)
			group_loops<2, FULL, NT>(w, T_dyn, tab, Gc, ga.prio, false);
		else
			group_loops<3, FULL, NT>(w, T_dyn, tab, Gc, ga.prio, build);
		if (kind == 3) { // (no sums; the waves that had nothing to do must not rebuild the table under the folding ones)
			__syncthreads();
			continue;
		}

		// ---- parity words -> the job's two sums -> its accumulator slots
		__syncthreads();
		// column n' = 32 h + n of matrix (pr, s) as a GF(2^64) element (bit p = G[p][n']); z = sum_n' col * e_n'  (gram.hpp tail_finish)
		for (unsigned task = wave; task < 6; task += kThreads / 64) {
			const unsigned pr = task / 3, s = task - 3 * pr;
			const unsigned h = lane >> 5, nn = lane & 31;
			auto spread = [](uint32_t x) { return (x & 0xFu) | ((x & 0xF0u) << 4) | ((x & 0xF00u) << 8) | ((x & 0xF000u) << 12); };
			const uint32_t *g0 = Gc[2 * pr + h][2 * s], *g1 = Gc[2 * pr + h][2 * s + 1];
			const uint32_t lo = spread(g0[nn]) | (spread(g0[nn + 32]) << 4);
			const uint32_t hi = spread(g1[nn]) | (spread(g1[nn + 32]) << 4);
			uint64_t z = mul_basis64((uint64_t)lo | ((uint64_t)hi << 32), lane);
#pragma unroll
			for (int mm = 32; mm >= 1; mm >>= 1)
				z ^= __shfl_xor(z, mm, 64);
			if (lane == 0) z3[pr][s] = z;
		}
		__syncthreads();
		if (tid < 2) s_loc[tid] = kara64(z3[tid][0], z3[tid][1], z3[tid][2]);
		__syncthreads();
		if (tid < 4) {
			const uint64_t v = reinterpret_cast<const uint64_t *>(s_loc)[tid];
			if (v) atomicXor(reinterpret_cast<unsigned long long *>(ga.S + jb->slot) + tid, (unsigned long long)v);
		}
	}
	// ---- the last workgroup of the launch publishes ALL slots
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (tid == 0) {
		const unsigned tk = atomicAdd(ga.counter, 1u);
		is_last = (tk == gridDim.x - 1) ? 1u : 0u;
	}
	__syncthreads();
	if (is_last) {
		// (everything below happens in wave 0, whose drain orders the value stores before the sequence word -- posted writes to one
		// destination keep their order; finalize.hpp)
		if (tid < 64) {
			for (unsigned i = tid; i < ga.n_slots; i += 64) {
				f128 v;
				v.lo = __hip_atomic_load(&ga.S[i].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				v.hi = __hip_atomic_load(&ga.S[i].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(&ga.vals[i].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(&ga.vals[i].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(&ga.S[i].lo, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(&ga.S[i].hi, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		if (tid == 0) {
			__hip_atomic_store(&ga.mail[64].lo, ga.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(ga.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}

// Deals the workgroups of one launch out to the UNITS of the job list -- a job with its chain (jobs_in[i].chain followers directly
// behind it), or a lone job -- in proportion to their traffic, eight at a time where a unit gets at least eight so that its tiles
// keep the XCD-aware order; sorts the table by first workgroup and launches.  jobs[i].slot is the caller's; wg_begin / wg_count are
// filled in here.  Every job: n >= 1; all jobs of a chain: the same n; kind 0 / 3 / 4 jobs write out[] (2 n elements each).
//
// PACKED launches.  With more units than a launch has workgroups to give each a sensible share (a prover of the keccak width: a
// hundred claims and more, piop/prove.rs:262-287), the units are packed into U super-units of n_cu / U workgroups each, U a power
// of two: every workgroup of a super-unit runs ALL its jobs one after the other, each job on the workgroup's share of that
// job's tiles -- the chain mechanism of the kernel, used for jobs that do not depend on each other (chains that do stay whole
// and in order inside one super-unit).  Against one unit per job this costs a parity reduction per job and workgroup (~2 us) and
// buys (1) any number of jobs per launch, (2) balance -- 100 equal jobs on 256 workgroups are 2.56 workgroups each, i.e. 78 % of the
// chip busy; as 4 super-units of 25 jobs on 64 workgroups they are 100 % --, (3) the XCD-aware tile order (ranges in multiples of
// eight), and (4) every workgroup of the chip walks the jobs in the same order, so that an array shared by many claims (a
// transparent against every committed column) is re-read while it is still in the Infinity Cache.  U is chosen by a cost
// model (time per pair of tiles by job kind at the measured per-CU rate, fixed cost per job) over the longest super-unit.
namespace {
struct pack_item {
	uint32_t first, len;
	double weight;
};
inline double pair_us(const group_job &j) // one pair of tiles (512 points) of this job on one workgroup at ~ 5 TB/s / 256 CUs: microseconds
{
	const bool one = j.kind == 3 && !j.x0[1];
	switch (j.kind) {
	case 0: return 5.0;              // 192 B per point
	case 3: return one ? 2.5 : 5.0;
	case 4: return 3.4;              // 128 B per point
	default: return 1.7;             // 64 B per point
	}
}
// what a further job costs a workgroup besides its tiles: the pipeline of loads, staging and Gram steps drains and refills, the
// parity words are reduced (measured: 50 equal jobs as 2 x 25 at 0.45 of the HBM peak against 0.52 for 64 jobs of 4 workgroups each)
inline double fixed_us(const group_job &j) { return j.kind == 3 ? 3.0 : 10.0; }
} // namespace

hipError_t launch_group(hipStream_t s, int n_cu, const group_job *jobs_in, uint32_t n_jobs, uint32_t n_slots, f128 *d_S, f128 *d_vals, f128 *d_mail,
                        unsigned *d_counter, uint64_t seq, group_job *h_table, const group_job *d_table)
{
	if (n_jobs == 0 || n_jobs > (uint32_t)kGroupMaxJobs || n_slots > (uint32_t)kGroupMaxSlots || !h_table || !d_table) return hipErrorNotSupported;
	if (n_cu > kGroupMaxGrid) n_cu = kGroupMaxGrid;
	static const bool prof_launch = getenv("BN_GROUP_PROF") != nullptr; // (diagnostic: host time of this function by step, and of the runtime's launch call itself)
	static double prof_step_us[3] = {0, 0, 0};
	auto t_step = prof_launch ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point{};
	auto step_lap = [&](int i) {
		if (!prof_launch) return;
		const auto now = std::chrono::steady_clock::now();
		prof_step_us[i] += std::chrono::duration<double, std::micro>(now - t_step).count();
		t_step = now;
	};
	static const uint32_t prio = [] {
		const char *e = bn::settled_knob("BN_FE_FP4_PRIO");
		return e ? (uint32_t)atoi(e) & 3u : 3u;
	}();
	static group_kargs ga_zero{};
	group_kargs ga = ga_zero;
	bool full = true;
	uint64_t total_elems = 0;
	std::vector<double> w;
	std::vector<uint32_t> tiles, cap, cnt, first, len;
	w.reserve(n_jobs);
	tiles.reserve(n_jobs);
	cap.reserve(n_jobs);
	first.reserve(n_jobs);
	len.reserve(n_jobs);
	double W = 0;
	uint32_t n_units = 0, max_tiles = 0;
	for (uint32_t i = 0; i < n_jobs;) {
		const uint32_t l = 1 + jobs_in[i].chain;
		if (i + l > n_jobs) return hipErrorInvalidValue;
		n_units++;
		first.push_back(i);
		len.push_back(l);
		double wu = 0;
		const uint64_t nt = (jobs_in[i].n + kTP - 1) / kTP;
		for (uint32_t q = i; q < i + l; q++) {
			const group_job &j = jobs_in[q];
			if (j.n == 0 || j.kind > 4 || (j.kind != 3 && j.slot + 2 > n_slots) || j.n != jobs_in[i].n || (q > i && j.chain)) return hipErrorInvalidValue;
			if (q == i && j.acquire) return hipErrorInvalidValue; // (nothing in front of it to read back)
			if (j.n % kTP) full = false;
			const bool one = j.kind == 3 && !j.x0[1];
			wu += (double)nt * (j.kind == 0 ? 3.0 : j.kind == 3 ? (one ? 1.5 : 3.0) : j.kind == 4 ? 2.0 : 1.0);
			total_elems += j.n * (j.kind == 0 ? 8 : j.kind == 3 ? (one ? 3 : 6) : j.kind == 4 ? 6 : 4);
		}
		if (nt > (1ull << 14) * (uint64_t)n_cu) return hipErrorNotSupported; // 2^22 points per workgroup and job: the f32 counts stay exact
		w.push_back(wu);
		tiles.push_back((uint32_t)nt);
		cap.push_back((uint32_t)(nt >= 2 ? nt / 2 : 1)); // a workgroup wants a pair of tiles
		if ((uint32_t)nt > max_tiles) max_tiles = (uint32_t)nt;
		W += wu;
		i += l;
	}
	uint32_t at = 0;
	bool packed = n_units > 32 || (int)n_units > n_cu;
	// (one unit per job where that is no worse by the same model: every unit its proportional share of the workgroups)
	double unpacked_cost = -1;
	if (packed && (int)n_units <= n_cu) { // (cheap: one pass over the jobs)
		unpacked_cost = 0;
		for (uint32_t i = 0; i < n_units; i++) {
			uint32_t c = (uint32_t)((double)n_cu * w[i] / W);
			if (c < 1) c = 1;
			if (c > cap[i]) c = cap[i];
			double cu = 0;
			for (uint32_t q = first[i]; q < first[i] + len[i]; q++) cu += (double)(((uint64_t)tiles[i] + 2 * c - 1) / (2 * c)) * pair_us(jobs_in[q]) + fixed_us(jobs_in[q]);
			if (cu > unpacked_cost) unpacked_cost = cu;
		}
	}
	uint32_t best_U = 0;
	double best_cost = 0;
	std::vector<uint32_t> best_assign;
	// (a prover's rounds repeat one shape with halving sizes: the decision for "these many units of this much weight on this many
	// tiles" is remembered -- the model costs ~10 us of host time per launch at a hundred units)
	struct decision {
		uint32_t n_units = 0, max_tiles = 0, U = 0;
		int n_cu = 0;
		double W = 0;
		bool packed = false;
		std::vector<uint32_t> assign;
	};
	// (one entry per round size: a prove walks max_tiles down by halves, the next prove of the shape finds every one of them)
	static thread_local std::vector<decision> remembered_decisions(64);
	decision &last = remembered_decisions[(n_units * 31u + max_tiles * 7u + (uint32_t)n_cu) & 63u];
	static const bool pack_forced = bn::settled_knob("BN_GROUP_PACK_U") != nullptr;
	const bool remembered = packed && last.n_units == n_units && last.max_tiles == max_tiles && last.W == W && last.n_cu == n_cu && !pack_forced;
	if (remembered) {
		packed = last.packed;
		best_U = last.U;
		best_assign = last.assign;
	}
	if (packed && !remembered) {
		// ---- U super-units of g = n_cu / U workgroups; items (units) dealt out heaviest first to the least loaded super-unit
		std::vector<uint32_t> by_weight(n_units);
		for (uint32_t i = 0; i < n_units; i++) by_weight[i] = i;
		std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
		const uint32_t g_min = (max_tiles + (1u << 14) - 1) >> 14; // the exactness bound
		std::vector<uint32_t> assign(n_units);
		for (uint32_t U = 1; U <= (uint32_t)n_cu && U <= n_units; U <<= 1) {
			const uint32_t g = (uint32_t)n_cu / U;
			if (g < g_min || g == 0) break;
			std::vector<double> cost(U, 0.0);
			// (least loaded super-unit first: a heap -- a constraint set's request is hundreds of units)
			typedef std::pair<double, uint32_t> lu;
			std::priority_queue<lu, std::vector<lu>, std::greater<lu>> heap;
			for (uint32_t q = 0; q < U; q++) heap.push(lu{0.0, q});
			for (uint32_t o = 0; o < n_units; o++) {
				const uint32_t it = by_weight[o];
				const lu top = heap.top();
				heap.pop();
				const uint32_t u = top.second;
				assign[it] = u;
				heap.push(lu{top.first + w[it], u});
				for (uint32_t q = first[it]; q < first[it] + len[it]; q++) {
					const uint64_t steps = ((uint64_t)tiles[it] + 2 * g - 1) / (2 * g);
					cost[u] += (double)steps * pair_us(jobs_in[q]) + fixed_us(jobs_in[q]);
				}
			}
			double c = 0;
			for (uint32_t u = 0; u < U; u++)
				if (cost[u] > c) c = cost[u];
			if (!best_U || c < best_cost) {
				best_U = U;
				best_cost = c;
				best_assign = assign;
			}
		}
		if (!best_U && unpacked_cost < 0) return hipErrorNotSupported;
		if (unpacked_cost >= 0 && (!best_U || unpacked_cost <= best_cost)) packed = false;
		static const int force_u = [] { // (BN_GROUP_PACK_U, measurement builds: that many super-units; -1: one unit per job -- tools/r06_pack_sweep.sh)
			const char *e = bn::settled_knob("BN_GROUP_PACK_U");
			return e ? atoi(e) : 0;
		}();
		if (force_u < 0 && unpacked_cost >= 0) packed = false;
		if (force_u > 0 && (uint32_t)force_u <= n_units && (uint32_t)n_cu / (uint32_t)force_u >= (g_min ? g_min : 1)) {
			packed = true;
			best_U = (uint32_t)force_u;
			std::vector<double> load(best_U, 0.0);
			best_assign.assign(n_units, 0);
			for (uint32_t o = 0; o < n_units; o++) {
				const uint32_t it = by_weight[o];
				uint32_t u = 0;
				for (uint32_t q = 1; q < best_U; q++)
					if (load[q] < load[u]) u = q;
				best_assign[it] = u;
				load[u] += w[it];
			}
		}
	}
	if (!remembered && (n_units > 32 || (int)n_units > n_cu)) {
		last.n_units = n_units;
		last.n_cu = n_cu;
		last.max_tiles = max_tiles;
		last.W = W;
		last.packed = packed;
		last.U = best_U;
		last.assign = best_assign;
	}
	step_lap(0);
	static thread_local std::vector<group_job> staged;
	if (staged.size() < n_jobs) staged.resize(n_jobs);
	if (packed) {
		const uint32_t U = best_U, g = (uint32_t)n_cu / U;
		// the items of every super-unit, in the caller's order (a chain's jobs stay in order): one counting pass -- a scan of all items
		// per super-unit was 20 us of host time per launch at three hundred jobs on 128 super-units
		std::vector<uint32_t> u_begin(U + 1, 0), by_unit(n_units);
		for (uint32_t it = 0; it < n_units; it++) u_begin[best_assign[it] + 1]++;
		for (uint32_t u = 0; u < U; u++) u_begin[u + 1] += u_begin[u];
		{
			std::vector<uint32_t> fill(u_begin.begin(), u_begin.end() - 1);
			for (uint32_t it = 0; it < n_units; it++) by_unit[fill[best_assign[it]]++] = it;
		}
		uint32_t k = 0;
		for (uint32_t u = 0; u < U; u++) {
			const uint32_t head = k;
			for (uint32_t o = u_begin[u]; o < u_begin[u + 1]; o++) {
				const uint32_t it = by_unit[o];
				for (uint32_t q = 0; q < len[it]; q++, k++) {
					staged[k] = jobs_in[first[it] + q];
					staged[k].wg_begin = u * g;
					staged[k].wg_count = 0;
					staged[k].chain = 0;
				}
			}
			if (k == head) continue; // (more super-units than items cannot happen: U <= n_units and the heaviest-first deal fills every one)
			staged[head].wg_count = g;
			staged[head].chain = k - head - 1;
			for (uint32_t x = u * g; x < (u + 1) * g; x++) ga.head_of_wg[x >> 1] |= head << (16 * (x & 1));
		}
		at = U * g;
	} else {
		if (n_cu < (int)n_units) return hipErrorNotSupported;
		// the workgroups of every unit: its proportional share -- in multiples of eight where that costs nothing (the XCD-aware tile order
		// wants ranges in multiples of eight), plainly where the rounding would leave the launch out of balance: twelve equal jobs on
		// 256 workgroups are 21.3 each; as 8 x 24 + 4 x 16 the launch lasts as long as its 16-workgroup jobs (measured, k = 12 / n = 24:
		// the fused launches at 0.47 - 0.50 of the HBM peak against 0.58 for k = 3 and 0.55 for k = 50), as 4 x 22 + 8 x 21 it is
		// balanced to 1.6 % (0.56)
		auto deal = [&](bool round8, std::vector<uint32_t> &c_out, double &max_load) -> bool {
			c_out.assign(n_units, 0);
			// first pass: the proportional share, rounded down (round8: to a multiple of eight from eight on), at least the workgroups
			// the exactness bound asks for, at most one per pair of tiles
			uint32_t used = 0;
			for (uint32_t i = 0; i < n_units; i++) {
				uint32_t c = (uint32_t)((double)n_cu * w[i] / W);
				if (round8 && c >= 8) c &= ~7u;
				const uint32_t need = (tiles[i] + (1u << 14) - 1) >> 14;
				if (c < need) c = need;
				if (c > cap[i]) c = cap[i];
				if (c < 1) c = 1;
				c_out[i] = c;
				used += c;
			}
			// (minimums can overshoot only with very many very uneven units: take from the largest)
			while (used > (uint32_t)n_cu) {
				uint32_t big = 0;
				for (uint32_t i = 1; i < n_units; i++)
					if (c_out[i] > c_out[big]) big = i;
				if (c_out[big] <= 1) return false;
				const uint32_t need = (tiles[big] + (1u << 14) - 1) >> 14;
				if (c_out[big] - 1 < need) return false;
				c_out[big]--;
				used--;
			}
			// the rest goes to whoever has the most work per workgroup (round8: eight at a time for units in multiples of eight)
			// (a heap: the heaviest load first, the lower index among equals -- a unit that cannot grow any more never can again, counts
			// and `used` only rise; the scan of all units per workgroup handed out was 25 us of host time per launch at 175 units)
			{
				typedef std::pair<double, uint32_t> lu; // (load, n_units - 1 - index)
				std::priority_queue<lu> heap;
				for (uint32_t i = 0; i < n_units; i++)
					if (w[i] > 0) heap.push(lu{w[i] / c_out[i], n_units - 1 - i});
				while (!heap.empty() && used < (uint32_t)n_cu) {
					const uint32_t i = n_units - 1 - heap.top().second;
					heap.pop();
					const uint32_t step = (round8 && c_out[i] >= 8 && (c_out[i] & 7) == 0) ? 8 : 1;
					if (c_out[i] + step > cap[i] || used + step > (uint32_t)n_cu) continue;
					c_out[i] += step;
					used += step;
					heap.push(lu{w[i] / c_out[i], n_units - 1 - i});
				}
			}
			max_load = 0;
			for (uint32_t i = 0; i < n_units; i++)
				if (w[i] / c_out[i] > max_load) max_load = w[i] / c_out[i];
			return true;
		};
		double load8 = 0, load1 = 0;
		std::vector<uint32_t> cnt1;
		const bool ok8 = deal(true, cnt, load8), ok1 = n_units > 1 && deal(false, cnt1, load1);
		if (!ok8 && !ok1) return hipErrorNotSupported;
		if (ok1 && (!ok8 || load1 < 0.97 * load8)) cnt.swap(cnt1); // (the XCD-aware order is worth a few per cent, measured on the single-claim kernels)
		// table order: the units whose count is a multiple of eight first (their ranges then start on multiples of eight)
		std::vector<uint32_t> order;
		for (uint32_t i = 0; i < n_units; i++)
			if ((cnt[i] & 7) == 0) order.push_back(i);
		for (uint32_t i = 0; i < n_units; i++)
			if ((cnt[i] & 7) != 0) order.push_back(i);
		uint32_t k = 0;
		for (uint32_t o = 0; o < n_units; o++) {
			const uint32_t u = order[o];
			for (uint32_t g = at; g < at + cnt[u]; g++) ga.head_of_wg[g >> 1] |= k << (16 * (g & 1));
			for (uint32_t q = 0; q < len[u]; q++, k++) {
				staged[k] = jobs_in[first[u] + q];
				staged[k].wg_begin = at;
				staged[k].wg_count = q == 0 ? cnt[u] : 0;
			}
			at += cnt[u];
		}
	}
	// (the table is put together in ordinary memory and goes to the pinned block in one sequential copy: the block is written once,
	// front to back, never read or patched by the host)
	step_lap(1);
	std::memcpy(h_table, staged.data(), (size_t)n_jobs * sizeof(group_job));
	__atomic_thread_fence(__ATOMIC_SEQ_CST); // (the table is in memory before the doorbell rings)
	step_lap(2);
	ga.table = d_table;
	ga.S = d_S;
	ga.vals = d_vals;
	ga.mail = d_mail;
	ga.counter = d_counter;
	ga.seq = seq;
	ga.n_jobs = n_jobs;
	ga.n_slots = n_slots;
	static const uint32_t adj_tiles = [] {
		const char *e = settled_knob("BN_GROUP_ADJ"); // (round-5 A/B, profiles/DEAD_ENDS.md: adjacent tiles per pair of fold groups -- no difference)
		return e ? (uint32_t)(atoi(e) != 0) : 0u;
	}();
	ga.prio = prio | (adj_tiles << 2);
	static const int nt_min_log2 = [] {
		const char *e = bn::settled_knob("BN_FE_NT_MIN_LOG2");
		return e ? atoi(e) : 25;
	}();
	// streaming accesses once the launch's arrays cannot stay in the caches anyway (the single-claim kernels' threshold is 2^25
	// elements per array = 2^27 elements touched)
	// (BN_GROUP_NT_MIN_LOG2, measurement builds: the same threshold for launches of the group kernel only; A/B of round 6 within the
	// noise in both directions, profiles/r06/README.md)
	static const int nt_group_log2 = [] {
		const char *e = bn::settled_knob("BN_GROUP_NT_MIN_LOG2");
		return e ? atoi(e) : -1;
	}();
	const int nt_log2 = nt_group_log2 >= 0 ? nt_group_log2 : nt_min_log2;
	const bool nt = full && nt_log2 < 62 && total_elems >= (4ull << nt_log2);
	constexpr unsigned lds = 2 * kFoldGroups * kTile4W * 4;
	{
		const void *fn[4] = {reinterpret_cast<const void *>(&k_group_fp4<false, false>), reinterpret_cast<const void *>(&k_group_fp4<true, false>),
		                     reinterpret_cast<const void *>(&k_group_fp4<true, true>), nullptr};
		for (int i = 0; fn[i]; i++) {
			const hipError_t e = func_lds_limit(fn[i], lds);
			if (e != hipSuccess) return e;
		}
	}
	const auto t_l0 = prof_launch ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point{};
	struct launch_timer {
		bool on;
		std::chrono::steady_clock::time_point t0;
		~launch_timer()
		{
			if (!on) return;
			static uint64_t ns = 0, calls = 0;
			ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
			if ((++calls & 63) == 0)
				fprintf(stderr, "[bn group prof] hipLaunchKernel of the group kernel: %.2f us per call over %llu calls; in front of it, us per call: weights + decision %.2f, dealing %.2f, table copy %.2f\n",
				        ns / 1e3 / calls, (unsigned long long)calls, prof_step_us[0] / calls, prof_step_us[1] / calls, prof_step_us[2] / calls);
		}
	} lt{prof_launch, t_l0};
	if (nt)
		hipLaunchKernelGGL((k_group_fp4<true, true>), dim3(at), dim3(kThreads), lds, s, ga);
	else if (full)
		hipLaunchKernelGGL((k_group_fp4<true, false>), dim3(at), dim3(kThreads), lds, s, ga);
	else
		hipLaunchKernelGGL((k_group_fp4<false, false>), dim3(at), dim3(kThreads), lds, s, ga);
	return hipGetLastError();
}

// ---- host tail of a claim group (abi_group.cpp "hosted sessions") ------------------------------------------------------------
// Hand-over: every array of a prover (count arrays of n elements: at most kGroupTailMaxElems elements in all) goes to the host in the
// power basis of hostmul_clmul.cpp -- after the prover's deferred fold, performed here on the way (fold[j]: y = src0 + z (src0 +
// x1), stored to out[j] as the fold asked) or as it is (x1[j] == null: y = src0[j][i], nothing stored).  Phi is GF(2)-linear:
// one nibble-table product per element (table in device memory, ctable.hpp layout).  The staging is written with plain posted
// stores and validates itself: the last workgroup publishes the XOR over all elements of mirror_mix(value, index) XOR seq in
// mailbox word 66, then the sequence word; the host accepts the staging only when the tag it computes from what it reads is
// that tag (the protocol of kernels_foldeval8.hip's hand-over).
namespace {
__device__ __forceinline__ uint64_t gt_mix(uint64_t lo, uint64_t hi, uint64_t idx)
{
	const unsigned r1 = (unsigned)(idx & 63), r2 = (unsigned)((idx * 7 + 17) & 63);
	return ((lo << r1) | (r1 ? lo >> (64 - r1) : 0)) ^ ((hi << r2) | (r2 ? hi >> (64 - r2) : 0)) ^ (idx + 1) * 0x9E3779B97F4A7C15ull;
}
} // namespace

// (the pointers of the arrays a workgroup touches -- consecutive ones, at most 256 -- come from the pinned pointer table into shared
// memory once per workgroup)
__global__ __launch_bounds__(256) void k_group_mirror(group_mirror_args a)
{
	__shared__ ctable_smem tab;
	__shared__ uint4 phi_T[512];
	__shared__ uint64_t wtag[4];
	__shared__ unsigned is_last;
	__shared__ const void *p_src0[256], *p_x1[256];
	__shared__ void *p_out[256];
	const unsigned tid = threadIdx.x;
	const uint64_t g0 = (uint64_t)blockIdx.x * 256, g = g0 + tid, total = (uint64_t)a.count * a.n;
	const bool act = g < total;
	const uint32_t arr0 = (uint32_t)(g0 / a.n);
	const uint32_t arr = act ? (uint32_t)(g / a.n) : arr0, i = act ? (uint32_t)(g - (uint64_t)arr * a.n) : 0;
	if (arr0 + tid < a.count && (uint64_t)(arr0 + tid) * a.n < g0 + 256) {
		p_src0[tid] = a.ptrs->src0[arr0 + tid];
		p_x1[tid] = a.ptrs->x1[arr0 + tid];
		p_out[tid] = a.ptrs->out[arr0 + tid];
	}
	phi_T[tid] = a.phi_tab[tid];
	phi_T[tid + 256] = a.phi_tab[tid + 256];
	ctable_build(tab, a.z); // (ends with a barrier: phi_T and the pointers are complete too)
	const uint4 *src0 = reinterpret_cast<const uint4 *>(p_src0[arr - arr0]), *x1 = reinterpret_cast<const uint4 *>(p_x1[arr - arr0]);
	uint4 v0{0, 0, 0, 0}, v1{0, 0, 0, 0};
	if (act) {
		v0 = src0[i];
		if (x1) v1 = x1[i];
	}
	uint64_t tag = 0;
	if (act) {
		uint4 y = v0;
		if (x1) {
			y = xor4(v0, ctable_mul(tab, xor4(v0, v1)));
			reinterpret_cast<uint4 *>(p_out[arr - arr0])[i] = y;
		}
		const uint4 py = ctable_mul(*reinterpret_cast<const ctable_smem *>(phi_T), y);
		reinterpret_cast<uint4 *>(a.staging)[g] = py;
		tag = gt_mix((uint64_t)py.x | ((uint64_t)py.y << 32), (uint64_t)py.z | ((uint64_t)py.w << 32), g);
	}
#pragma unroll
	for (int sh = 32; sh >= 1; sh >>= 1) tag ^= __shfl_xor(tag, sh, 64);
	if ((tid & 63) == 0) wtag[tid >> 6] = tag;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (tid == 0) {
		const uint64_t t = wtag[0] ^ wtag[1] ^ wtag[2] ^ wtag[3];
		if (t) atomicXor(reinterpret_cast<unsigned long long *>(a.tag_acc), (unsigned long long)t);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		const unsigned tk = atomicAdd(a.counter, 1u);
		is_last = tk == gridDim.x - 1 ? 1u : 0u;
		if (is_last) {
			const uint64_t all = __hip_atomic_load(a.tag_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(a.tag_acc, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&a.mail[66].lo, all ^ a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__hip_atomic_store(&a.mail[64].lo, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

hipError_t launch_group_mirror(hipStream_t s, const group_mirror_args &a)
{
	if (a.count == 0 || a.count > (uint32_t)kGroupMaxArrays || a.n == 0 || (uint64_t)a.count * a.n > kGroupTailMaxElems || !a.ptrs || !a.staging || !a.phi_tab ||
	    !a.tag_acc || !a.counter || !a.mail)
		return hipErrorNotSupported;
	__atomic_thread_fence(__ATOMIC_SEQ_CST); // (the pointer table is in memory before the doorbell rings)
	hipLaunchKernelGGL(k_group_mirror, dim3((unsigned)(((uint64_t)a.count * a.n + 255) / 256)), dim3(256), 0, s, a);
	return hipGetLastError();
}

// The device catching up with a hosted prover: out[j][i] = PhiInv(staging[j * n0 + i]), i < n0 -- the first n0 elements of the
// host's folded copies, which are exactly what the folds the caller asked for leave in its buffers (kernels_stream.hip
// k_tail_writeback for any number of arrays).
__global__ __launch_bounds__(256) void k_group_writeback(group_writeback_args a)
{
	__shared__ uint4 T[512];
	__shared__ void *p_out[256];
	const unsigned tid = threadIdx.x;
	const uint64_t g0 = (uint64_t)blockIdx.x * 256, g = g0 + tid, total = (uint64_t)a.count * a.n0;
	const bool act = g < total;
	const uint32_t arr0 = (uint32_t)(g0 / a.n0);
	const uint32_t arr = act ? (uint32_t)(g / a.n0) : arr0, i = act ? (uint32_t)(g - (uint64_t)arr * a.n0) : 0;
	if (arr0 + tid < a.count && (uint64_t)(arr0 + tid) * a.n0 < g0 + 256) p_out[tid] = a.ptrs->out[arr0 + tid];
	uint4 v{0, 0, 0, 0};
	if (act) v = reinterpret_cast<const uint4 *>(a.staging)[g];
	T[tid] = a.phi_inv[tid];
	T[tid + 256] = a.phi_inv[tid + 256];
	__syncthreads();
	if (act) reinterpret_cast<uint4 *>(p_out[arr - arr0])[i] = ctable_mul(*reinterpret_cast<const ctable_smem *>(T), v);
}

hipError_t launch_group_writeback(hipStream_t s, const group_writeback_args &a)
{
	if (a.count == 0 || a.count > (uint32_t)kGroupMaxArrays || a.n0 == 0 || (uint64_t)a.count * a.n0 > kGroupTailMaxElems || !a.ptrs || !a.staging || !a.phi_inv)
		return hipErrorNotSupported;
	__atomic_thread_fence(__ATOMIC_SEQ_CST);
	hipLaunchKernelGGL(k_group_writeback, dim3((unsigned)(((uint64_t)a.count * a.n0 + 255) / 256)), dim3(256), 0, s, a);
	return hipGetLastError();
}

} // namespace bn
