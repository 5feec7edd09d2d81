// binius_amd/csrc/kernels_foldeval_fp4.hip -- the fused fold + round evaluation of kernels_foldeval_mfma.hip with the
// two halves of the work on DIFFERENT waves and the Gram products on the FP4 matrix path (gram_fp4.hpp):
//
//   a'[i] = a[i] + z*(a[i + N/2] - a[i])        i < N/2        (extrapolate_line, layer.rs:421)
//   S_1   = sum_{j < N/4} a'[j + N/4] * b'[j + N/4]
//   S_inf = sum_{j < N/4} (a'[j] + a'[j + N/4]) * (b'[j] + b'[j + N/4])
//                                                (v3/bivariate_product.rs:217-228 then :303-408)
//
// Why a second form.  In kernels_foldeval_mfma.hip every wave does both jobs: per tile of 256 points ~810 VALU
// instructions and 176 LDS reads of the constant multiplication plus 48 int8 MFMAs, and the measured issue model of this
// chip (tools/mfma_issue.hip, tools/mfma_issue_fp4.hip: an MFMA 32x32 costs a SIMD 18 - 20 ns back to back on real
// operands, 4 - 5 VALU instructions ride along, every further one adds 1.2 ns) puts that kernel at VALU + MFMA issue,
// not at memory.  The FP4 path needs 24 MFMAs per tile instead of 48, but its k-steps next to the
// constant multiplication in ONE wave did not fit the register file (61 - 85 spilled registers).  Here
// a workgroup is 12 waves on one CU: waves 4 .. 11 ("fold waves", two per SIMD) load, fold, store and stage a pair of
// tiles into T[i & 1] while waves 0 .. 3 ("Gram waves", one per SIMD) run the FP4 k-steps of the previous pair out of
// T[(i - 1) & 1] -- each kind of wave has 168 registers for its own job alone, a SIMD's issue slots go to two VALU
// streams and one MFMA stream, and one workgroup barrier per pair hands the tiles over.
//
// Whole tiles, at least two per CU; everything else stays with kernels_foldeval_mfma.hip.
// Algorithmic bytes: read 16*m*N + write 8*m*N = 24*m*N per launch.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "arm.hpp"
#include "ctable.hpp"
#include "gram_fp4.hpp"

namespace bn {

using namespace gram4;

namespace {

typedef unsigned int fq_v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 fq_load(const uint4 *p)
{
	if constexpr (NT) {
		const fq_v4u v = __builtin_nontemporal_load(reinterpret_cast<const fq_v4u *>(p));
		return uint4{v.x, v.y, v.z, v.w};
	} else {
		return *p;
	}
}
template <bool NT>
__device__ __forceinline__ void fq_store(uint4 *p, uint4 r)
{
	if constexpr (NT) {
		const fq_v4u v = {r.x, r.y, r.z, r.w};
		__builtin_nontemporal_store(v, reinterpret_cast<fq_v4u *>(p));
	} else {
		*p = r;
	}
}

constexpr unsigned kGramWaves = 4;  // waves 0 .. 3 run the Gram k-steps (one per SIMD)
constexpr unsigned kFoldGroups = 2; // waves 4 .. 11 fold: two groups of four waves (two fold waves per SIMD), a tile per group
constexpr unsigned kThreads = 64 * kGramWaves * (1 + kFoldGroups);

} // namespace

// One iteration of a workgroup = a PAIR of tiles: fold group g folds the pair's tile g into T[buf][g] while the Gram waves run
// the k-steps of the previous pair out of T[buf ^ 1][0 .. 1]; one workgroup barrier per pair.  (One fold wave per SIMD -- 512
// threads, a tile per iteration -- was built first and is latency-bound: a single wave issues a dependent VALU instruction every
// ~4.7 cycles and waits out every group of table reads alone; 0.546 of the HBM peak against 0.556 for kernels_foldeval_mfma.hip
// on the same box, profiles/r04/experiments/fe_fp4_v1.txt.)
// SC: 0 = plain fold; 1 / 2 = the upper half of folded array 0 / 1 is multiplied by fa.hi_scale (a fifth constant multiplication
// per point, through a second nibble table) before it is stored and staged (bn_extrapolate_line_batch_scaled).
template <int SC, bool NT>
__global__ __launch_bounds__(kThreads, 1) void k_foldeval_mfma_fp4(foldeval_args fa, uint64_t n_in, f128 z, f128 *out, fin_fuse fz, arm_args arm)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t T_dyn[]; // 2 buffers x 2 tiles of FP4 operands
	__shared__ ctable_smem tab;
	__shared__ ctable_opt<SC != 0> tab_hs;
	__shared__ fin_cache fcache;
	__shared__ gram_parity Gc;
	const uint64_t seq = fz.args.seq;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool folds = wave >= kGramWaves;
	const unsigned grp = folds ? (wave - kGramWaves) >> 2 : 0;   // fold group: which tile of the pair
	const unsigned ftid = (threadIdx.x - 64 * kGramWaves) & 255; // the lane's point inside its tile (fold waves)
	const uint64_t n = n_in >> 2;                                // evaluation points of the next round (a multiple of kTP)
	const uint64_t n_tiles = n / kTP;

	// tile order: see kernels_foldeval_mfma.hip (XCD x = blockIdx.x & 7 takes the x-th contiguous eighth of the tiles)
	uint64_t tbase = 0, tstride = gridDim.x, tlimit = n_tiles, t0 = blockIdx.x;
	if ((fa.xcd_tiles & 1) && (gridDim.x & 7) == 0) {
		const uint32_t chunk = (uint32_t)((n_tiles + 7) >> 3);
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		tlimit = tbase >= (uint32_t)n_tiles ? 0 : ((uint32_t)n_tiles - tbase < chunk ? (uint32_t)n_tiles - tbase : chunk);
	}

	// quadrant k = 2 * array + half: element index half * n + point
	uint4 x0[4], x1[4];
	// addresses = a uniform 64-bit base (array, half, tile: scalar registers and scalar arithmetic) + the lane's constant 32-bit byte
	// offset: the loads and stores take the scalar-base form and the loop has no 64-bit vector address arithmetic (13 v_lshl_add_u64
	// per tile and fold wave otherwise)
	const uint32_t voff = ftid * 16u;
	// (the empty asm keeps the compiler from folding the lane offset into a loop-invariant 64-bit vector base, which would bring the
	// vector additions back)
	auto lane_off = [&]() {
		uint32_t v = voff;
		asm volatile("" : "+v"(v));
		return v;
	};
	uint32_t vo = lane_off();
	auto load1 = [&](uint32_t t, int k) {
		const uint64_t e = ((k & 1 ? n : 0) + (uint64_t)(tbase + t) * kTP) * 16; // (uniform)
		x0[k] = fq_load<NT>(reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(fa.x0[k >> 1]) + e + vo));
		x1[k] = fq_load<NT>(reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(fa.x1[k >> 1]) + e + vo));
	};
	const uint32_t tm0 = t0 + grp * tstride; // this fold group's first tile
	if (folds && tm0 < tlimit) {
#pragma unroll
		for (int k = 0; k < 4; k++)
			load1(tm0, k);
	}
	{
		// the finalize arguments travel with the first tile and wait in LDS for the tail (finalize.hpp)
		const fin_pref fpre = fin_prefetch(fz);
		if (arm.h_cmd) { // (uniform) armed launch of a mid-size round: the challenge arrives through the command block (arm.hpp)
			f128 hs_in;
			if (!arm_wait(arm, z, hs_in)) return;
			fa.hi_scale = hs_in;
		}
		if constexpr (SC != 0) { // both tables side by side (two groups of 128 threads): a build is one dependent chain
			const unsigned g2 = threadIdx.x >> 7;
			ctable_build_group(g2 == 1 ? tab_hs.get() : tab, g2 == 1 ? fa.hi_scale : z, g2 < 2 ? (threadIdx.x & 127) : 512u, 128);
		} else {
			ctable_build(tab, z); // the loads above are in flight meanwhile; ends with a barrier
		}
		fin_commit(fz, fpre, fcache);
	}

	// The fold waves are the critical path of a pair (588 VALU instructions per tile against the Gram wave's 151 + 24 MFMAs): they issue
	// ahead of the Gram wave of their SIMD (fa.xcd_tiles bits 1 .. 2 = their priority; 1.5 - 2.5 % on the n = 24 ... 28 steps, any level above the Gram waves' does it: experiments/fe_fp4_prio.txt).
	if (folds) {
		switch ((fa.xcd_tiles >> 1) & 3) {
		case 1: __builtin_amdgcn_s_setprio(1); break;
		case 2: __builtin_amdgcn_s_setprio(2); break;
		case 3: __builtin_amdgcn_s_setprio(3); break;
		default: break;
		}
	}
	if (folds) {
		const stage4_role sr = make_stage4_role(ftid);
		uint32_t *Tn = T_dyn + grp * kTile4W;
		unsigned buf = 0;
		for (uint32_t t = t0; t < tlimit; t += 2 * tstride) {
			const uint32_t tm = t + grp * tstride;
			if (tm < tlimit) { // (uniform; false only for group 1 on an odd last pair)
				// the last iteration re-requests its own tile (cache hits) instead of branching around the loads
				const uint32_t tn = tm + 2 * tstride < tlimit ? tm + 2 * tstride : tm;
				vo = lane_off();
				const uint64_t pt16 = (uint64_t)(tbase + tm) * kTP * 16; // (uniform)
				uint4 f[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					f[k] = ctable_mul_acc<8, true>(tab, xor4(x0[k], x1[k]), x0[k]);
					load1(tn, k); // this quadrant of the group's next tile flies from here on
					if constexpr (SC != 0) {
						if (k == 2 * SC - 1) f[k] = ctable_mul_pinned<8>(tab_hs.get(), f[k]);
					}
				}
#pragma unroll
				for (int k = 0; k < 4; k++)
					fq_store<NT>(reinterpret_cast<uint4 *>(reinterpret_cast<char *>(fa.out[k >> 1]) + (k & 1 ? n * 16 : 0) + pt16 + vo), f[k]);
				// half 1 is the evaluation at 1, half 0 its partner: sets 0 / 1 = u, v at 1; sets 2 / 3 = u, v at infinity
				stage4_elem(Tn, sr, 0, f[1]);
				stage4_elem(Tn, sr, 2, xor4(f[1], f[0]));
				stage4_elem(Tn, sr, 1, f[3]);
				stage4_elem(Tn, sr, 3, xor4(f[3], f[2]));
			}
			__syncthreads(); // the pair is staged; the Gram waves are done with the buffer this wave writes next
			buf ^= 1;
			Tn = T_dyn + (buf * kFoldGroups + grp) * kTile4W;
		}
	} else {
		const gram4_role gr = make_gram4_role(wave, lane);
		v16f acc[kAccTiles];
		acc4_zero(acc);
		unsigned buf = 0;
		for (uint32_t t = t0; t < tlimit; t += 2 * tstride) {
			__syncthreads();
			const uint32_t *Tp = T_dyn + buf * kFoldGroups * kTile4W;
			gram4_tile(Tp, gr, acc);
			if (t + tstride < tlimit) gram4_tile(Tp + kTile4W, gr, acc);
			buf ^= 1;
		}
		parity4(acc, gr, wave, lane, Gc);
	}
	tail_finish(Gc, wave, lane, out, fz, seq, &fcache);
}

bool foldeval_fp4_applies(int n_cu, const foldeval_args &fa, uint64_t n_in)
{
	// BN_FE_FP4=0: off.  BN_FE_FP4_MIN_LOG2: elements per array from which this form takes the launch.
	static const int min_log2 = [] {
		const char *e = getenv("BN_FE_FP4");
		if (e && e[0] == '0') return 64;
		const char *m = bn::settled_knob("BN_FE_FP4_MIN_LOG2");
		return m ? atoi(m) : 0;
	}();
	if (min_log2 >= 64 || fa.scale_mask > 2 || n_in < 4 || (n_in & 3)) return false;
	const uint64_t n = n_in >> 2;
	if (n % kTP) return false;
	const uint64_t n_tiles = n / kTP;
	if (n_tiles < 2 * (uint64_t)n_cu) return false; // (a workgroup needs a pair of tiles to occupy its fold waves)
	if (n_in < (1ull << min_log2)) return false;
	return (n_tiles + n_cu - 1) / n_cu <= (1ull << 14); // 2^22 points per workgroup: the f32 counts stay exact
}

hipError_t launch_foldeval_fp4(hipStream_t s, int n_cu, const foldeval_args &fa_in, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse &fz,
                               const arm_args &arm, bool nt)
{
	// BN_FE_FP4_PRIO=0 .. 3: issue priority of the fold waves (default 3; the Gram waves stay at 0)
	static const uint32_t prio = [] {
		const char *e = bn::settled_knob("BN_FE_FP4_PRIO");
		return e ? (uint32_t)atoi(e) & 3u : 3u;
	}();
	foldeval_args fa = fa_in;
	fa.xcd_tiles = (fa.xcd_tiles & 1u) | (prio << 1);
	const uint64_t n_tiles = (n_in >> 2) / kTP;
	// BN_FE_FP4_GRID=g (tests): g workgroups instead of one per CU -- on 256 CUs every size the prover reaches gives every workgroup
	// an even number of tiles and the XCD-aware order; other grids exercise the odd last pair and the plain striding
	static const unsigned grid_override = [] {
		const char *e = getenv("BN_FE_FP4_GRID");
		return e ? (unsigned)atoi(e) : 0u;
	}();
	unsigned grid = (unsigned)(n_tiles < (uint64_t)n_cu ? n_tiles : (uint64_t)n_cu);
	if (grid_override && grid_override <= grid && (n_tiles + grid_override - 1) / grid_override <= (1ull << 14)) grid = grid_override;
	constexpr unsigned lds = 2 * kFoldGroups * kTile4W * 4;
	const hipError_t attr = [] { // (per device: func_lds_limit)
		const void *fn[6] = {reinterpret_cast<const void *>(&k_foldeval_mfma_fp4<0, false>), reinterpret_cast<const void *>(&k_foldeval_mfma_fp4<0, true>),
		                     reinterpret_cast<const void *>(&k_foldeval_mfma_fp4<1, false>), reinterpret_cast<const void *>(&k_foldeval_mfma_fp4<1, true>),
		                     reinterpret_cast<const void *>(&k_foldeval_mfma_fp4<2, false>), reinterpret_cast<const void *>(&k_foldeval_mfma_fp4<2, true>)};
		for (const void *f : fn) {
			const hipError_t e = func_lds_limit(f, lds);
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	}();
	if (attr != hipSuccess) return attr;
#define BN_FQ_LAUNCH(SC)                                                                                                       \
	if (nt)                                                                                                                    \
		hipLaunchKernelGGL((k_foldeval_mfma_fp4<SC, true>), dim3(grid), dim3(kThreads), lds, s, fa, n_in, z, d_out, fz, arm);  \
	else                                                                                                                       \
		hipLaunchKernelGGL((k_foldeval_mfma_fp4<SC, false>), dim3(grid), dim3(kThreads), lds, s, fa, n_in, z, d_out, fz, arm);
	switch (fa.scale_mask) {
	case 0: BN_FQ_LAUNCH(0) break;
	case 1: BN_FQ_LAUNCH(1) break;
	default: BN_FQ_LAUNCH(2) break;
	}
#undef BN_FQ_LAUNCH
	return hipGetLastError();
}

} // namespace bn
