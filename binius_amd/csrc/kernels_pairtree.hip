// binius_amd/csrc/kernels_pairtree.hip -- the small levels of pairwise_product_reduce
// (crates/compute/src/layer.rs:505-520, crates/compute/src/cpu/layer.rs `pairwise_product_reduce`: level r holds the
// products of adjacent pairs of level r - 1; every level is returned).
//
// The large levels are element-wise products at throughput (kernels_mul9.hip: bit-sliced, 224 elements per wave-batch,
// one dependent chain of ~2400 instructions per batch, ~10 us per level whatever its size).  From 2^15 elements down a
// level has too few products for that chain to pay, and the tree is walked here instead:
//
// * a workgroup owns S = 2^LOG_S adjacent elements and walks LOG_S levels of ITS subtree without leaving the CU
//   (levels meet in LDS, every level is also stored to the caller's round output); the grid-wide dependency is one
//   launch per LOG_S levels instead of one per level;
// * a product is computed by G lanes: lane j multiplies b by the j-th (128 / G)-bit limb of a -- a scalar of the
//   subfield T_K, K = 7 - log2 G, acting limb-wise on b (binary_field.rs:361-412), a walk over its 2^K bits --, moves
//   the partial product to its place with the multiplications by X_K .. X_6 the limb's index asks for, and the G partial
//   products are XORed across the lanes (DPP inside a row of 16, bpermute above).  G = 8 at the first level of a full
//   subtree (T = 4 S threads), doubling as the level empties (a level takes as many lanes per product as the workgroup has): the chain of a level is 400 - 800 instructions instead of
//   2400, and the last fifteen levels of the tree cost what three small kernels cost instead of fifteen launches.
#include <hip/hip_runtime.h>

#include "gf128.hpp"
#include "internal.hpp"

namespace bn {

namespace {

template <int K>
__device__ __forceinline__ f128 place_limb(f128 r, unsigned j)
{
	// r * 2^(j * 2^K): X_k for every bit k - K of j
	if constexpr (K <= 0) { if (j & (1u << (0 - K))) r = mulx<0>(r); }
	if constexpr (K <= 1) { if (j & (1u << (1 - K))) r = mulx<1>(r); }
	if constexpr (K <= 2) { if (j & (1u << (2 - K))) r = mulx<2>(r); }
	if constexpr (K <= 3) { if (j & (1u << (3 - K))) r = mulx<3>(r); }
	if constexpr (K <= 4) { if (j & (1u << (4 - K))) r = mulx<4>(r); }
	if constexpr (K <= 5) { if (j & (1u << (5 - K))) r = mulx<5>(r); }
	if constexpr (K <= 6) { if (j & (1u << (6 - K))) r = mulx<6>(r); }
	return r;
}

template <int CTRL>
__device__ __forceinline__ uint32_t xor_dpp(uint32_t x) { return x ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true); }

// XOR over the G lanes of a group (G a power of two, groups aligned); every lane ends up with the sum
template <int G>
__device__ __forceinline__ uint32_t group_xor(uint32_t x)
{
	if constexpr (G >= 2) x = xor_dpp<0xB1>(x);  // quad_perm [1,0,3,2]
	if constexpr (G >= 4) x = xor_dpp<0x4E>(x);  // quad_perm [2,3,0,1]
	if constexpr (G >= 8) x = xor_dpp<0x141>(x); // row_half_mirror: lane 7 - j (the other quad)
	if constexpr (G >= 16) x = xor_dpp<0x140>(x); // row_mirror: lane 15 - j (the other half row)
	if constexpr (G >= 32) x ^= (uint32_t)__shfl_xor((int)x, 16);
	if constexpr (G >= 64) x ^= (uint32_t)__shfl_xor((int)x, 32);
	return x;
}

// a * b by the G lanes of a group; j = this lane's index in the group; the product is valid in every lane
template <int G>
__device__ __forceinline__ f128 group_product(f128 a, f128 b, unsigned j)
{
	constexpr int K = G == 8 ? 4 : (G == 16 ? 3 : (G == 32 ? 2 : 1));
	constexpr unsigned W = 1u << K;
	const unsigned bit = j * W;
	const uint64_t word = (bit & 64) ? a.hi : a.lo;
	const uint64_t limb = (word >> (bit & 63)) & ((1ull << W) - 1);
	f128 r = mul_walk<K>(b, limb);
	r = place_limb<K>(r, j);
	uint32_t w0 = group_xor<G>((uint32_t)r.lo), w1 = group_xor<G>((uint32_t)(r.lo >> 32));
	uint32_t w2 = group_xor<G>((uint32_t)r.hi), w3 = group_xor<G>((uint32_t)(r.hi >> 32));
	return f128{(uint64_t)w0 | ((uint64_t)w1 << 32), (uint64_t)w2 | ((uint64_t)w3 << 32)};
}

// One level inside a workgroup: n_prod products of adjacent pairs of src, kept in LDS for the next level and stored to
// the caller's round output.
template <int G>
__device__ __forceinline__ void tree_level(const f128 *__restrict__ src, f128 *__restrict__ keep, f128 *__restrict__ gout, unsigned n_prod)
{
	const unsigned q = threadIdx.x / G, j = threadIdx.x % G;
	if (q < n_prod) { // (whole groups)
		const f128 p = group_product<G>(src[2 * q], src[2 * q + 1], j);
		if (j == 0) {
			keep[q] = p;
			gout[q] = p;
		}
	}
}

// as many lanes per product as the workgroup has for this level (8 .. 64)
template <unsigned T>
__device__ __forceinline__ void tree_level_any(const f128 *src, f128 *keep, f128 *gout, unsigned n_prod)
{
	if (n_prod * 64 <= T)
		tree_level<64>(src, keep, gout, n_prod);
	else if (n_prod * 32 <= T)
		tree_level<32>(src, keep, gout, n_prod);
	else if (n_prod * 16 <= T)
		tree_level<16>(src, keep, gout, n_prod);
	else
		tree_level<8>(src, keep, gout, n_prod);
}

} // namespace

// Workgroup b: the n = args.n_levels <= LOG_S levels of the subtree over elements [b << n, (b + 1) << n) of args.in;
// level l (1-based) goes to args.out[l - 1] + (b << (n - l)).  T = max(64, 4 S) threads.
// (All the stages in ONE launch -- the workgroup that stores the last root of a group takes a device-scope ticket and walks
// the group's subtree -- was measured: 31.1 us against 13.6 + 8.8 + 5.3 us for the three launches of a 2^15 tree; a
// hand-over through agent-scope stores, a ticket and agent-scope loads costs what a launch costs.  Not kept.)
template <int LOG_S>
__global__ __launch_bounds__((4 << LOG_S) < 64 ? 64 : (4 << LOG_S)) void k_pairtree(pairtree_args args)
{
	constexpr unsigned S = 1u << LOG_S;
	constexpr unsigned T = (4 << LOG_S) < 64 ? 64 : (4 << LOG_S);
	__shared__ f128 buf[2][S / 2];
	const unsigned b = blockIdx.x, n = args.n_levels;
	unsigned n_prod = 1u << (n - 1);
	// level 1 straight from global memory (the lanes of a group read the same two elements)
	tree_level_any<T>(args.in + ((uint64_t)b << n), buf[0], args.out[0] + (uint64_t)b * n_prod, n_prod);
	unsigned cur = 0;
	for (unsigned l = 2; l <= n; l++) {
		__syncthreads();
		n_prod >>= 1;
		tree_level_any<T>(buf[cur], buf[cur ^ 1], args.out[l - 1] + (uint64_t)b * n_prod, n_prod);
		cur ^= 1;
	}
}

hipError_t launch_pairtree(hipStream_t s, const pairtree_args &args, uint32_t log_s, uint64_t n_groups)
{
	if (n_groups == 0 || args.n_levels == 0 || args.n_levels > log_s || log_s > 8) return hipErrorInvalidValue;
	const dim3 grid((unsigned)n_groups);
	switch (log_s) {
	case 1: hipLaunchKernelGGL(k_pairtree<1>, grid, dim3(64), 0, s, args); break;
	case 2: hipLaunchKernelGGL(k_pairtree<2>, grid, dim3(64), 0, s, args); break;
	case 3: hipLaunchKernelGGL(k_pairtree<3>, grid, dim3(64), 0, s, args); break;
	case 4: hipLaunchKernelGGL(k_pairtree<4>, grid, dim3(64), 0, s, args); break;
	case 5: hipLaunchKernelGGL(k_pairtree<5>, grid, dim3(128), 0, s, args); break;
	case 6: hipLaunchKernelGGL(k_pairtree<6>, grid, dim3(256), 0, s, args); break;
	case 7: hipLaunchKernelGGL(k_pairtree<7>, grid, dim3(512), 0, s, args); break;
	default: hipLaunchKernelGGL(k_pairtree<8>, grid, dim3(1024), 0, s, args); break;
	}
	return hipGetLastError();
}

} // namespace bn
