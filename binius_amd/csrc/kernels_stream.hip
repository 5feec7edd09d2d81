// binius_amd/csrc/kernels_stream.hip -- HBM-streaming kernels: fill, add / add_assign,
// extrapolate_line (the sumcheck fold) and tensor_expand.
//
// All of them are bound by HBM bandwidth: 128-bit coalesced loads/stores (one uint4 per lane =
// 1 KiB per wave instruction), grid-stride loops sized to keep >= 8 waves per SIMD in flight, and
// the launch-constant multiplier handled by LDS nibble tables (ctable.hpp) so the ALU work per
// element (~100 VALU + 32 conflict-free ds_read_b128) stays well under the memory time.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "ctable.hpp"
#include "internal.hpp"

namespace bn {

static inline unsigned grid_for(uint64_t n_items, unsigned per_block, int n_cu, unsigned blocks_per_cu)
{
	uint64_t want = (n_items + per_block - 1) / per_block;
	uint64_t cap = (uint64_t)n_cu * blocks_per_cu;
	if (want < 1) want = 1;
	return (unsigned)(want < cap ? want : cap);
}

__global__ __launch_bounds__(256) void k_fill(uint4 *dst, uint64_t n, uint4 v)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		dst[i] = v;
}

__global__ __launch_bounds__(256) void k_add_assign(uint4 *dst, const uint4 *src, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		dst[i] = xor4(dst[i], src[i]);
}

__global__ __launch_bounds__(256) void k_add(uint4 *dst, const uint4 *s1, const uint4 *s2, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		dst[i] = xor4(s1[i], s2[i]);
}

// streaming (non-temporal) 16-byte accesses for arrays far larger than the caches
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld16(const uint4 *p)
{
	if constexpr (NT) {
		const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
		return uint4{v.x, v.y, v.z, v.w};
	} else {
		return *p;
	}
}
template <bool NT>
__device__ __forceinline__ void st16(uint4 *p, uint4 r)
{
	if constexpr (NT) {
		const v4u v = {r.x, r.y, r.z, r.w};
		__builtin_nontemporal_store(v, reinterpret_cast<v4u *>(p));
	} else {
		*p = r;
	}
}

// extrapolate_line: x0[i] += (x1[i] - x0[i]) * z          (crates/compute/src/layer.rs:421,
// semantics of crates/compute/src/cpu/layer.rs:393-408).  Algorithmic traffic: read 32 B, write
// 16 B per output element = 24 B per input element of the multilinear being folded.
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_extrapolate_line(uint4 *__restrict__ x0, const uint4 *__restrict__ x1, uint64_t n,
                                                          f128 z)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	// main body: U independent elements per iteration so U*2 16-byte loads are in flight per lane
	for (; i + (U - 1) * stride < n; i += U * stride) {
		uint4 a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			a[u] = ld16<NT>(&x0[i + u * stride]);
			b[u] = ld16<NT>(&x1[i + u * stride]);
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			uint4 p = ctable_mul(tab, xor4(a[u], b[u]));
			st16<NT>(&x0[i + u * stride], xor4(a[u], p));
		}
	}
	for (; i < n; i += stride) {
		uint4 a = x0[i], b = x1[i];
		x0[i] = xor4(a, ctable_mul(tab, xor4(a, b)));
	}
}

// the same over `count` (evals_0, evals_1) pairs of equal length in one launch: blockIdx.y picks the
// pair.  This is what an executor `map` scope over the multilinears of a fold becomes
// (v3/bivariate_product.rs:217-228) -- one launch per round instead of one per multilinear.
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_extrapolate_line_batch(fold_batch fb, uint64_t n, f128 z)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	// (s0 == x0 for the in-place fold; the first fold of a prover reads evals_0 where the caller left it and writes the
	// fresh buffer -- the "copy evals_0, then fold in place" of v3/bivariate_product.rs:196-206 in one pass)
	uint4 *x0 = (uint4 *)fb.x0[blockIdx.y];
	const uint4 *s0 = fb.src0[blockIdx.y] ? (const uint4 *)fb.src0[blockIdx.y] : (const uint4 *)x0;
	const uint4 *x1 = (const uint4 *)fb.x1[blockIdx.y];
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n; i += U * stride) {
		uint4 a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			a[u] = ld16<NT>(&s0[i + u * stride]);
			b[u] = ld16<NT>(&x1[i + u * stride]);
		}
#pragma unroll
		for (int u = 0; u < U; u++)
			st16<NT>(&x0[i + u * stride], xor4(a[u], ctable_mul(tab, xor4(a[u], b[u]))));
	}
	for (; i < n; i += stride) {
		uint4 a = s0[i], b = x1[i];
		x0[i] = xor4(a, ctable_mul(tab, xor4(a, b)));
	}
}

// The same fold for arrays of DIFFERENT lengths under one challenge: the folds of several provers of a batch round (one
// BivariateSumcheckProver per size, one challenge for all of them: prove/front_loaded.rs:122-155) in one launch.  blockIdx.y =
// the array; the grid's x extent is sized for the longest one.
template <int U>
__global__ __launch_bounds__(256) void k_extrapolate_line_ragged(fold_batch fb, fold_lengths fl, f128 z)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	uint4 *x0 = (uint4 *)fb.x0[blockIdx.y];
	const uint4 *s0 = fb.src0[blockIdx.y] ? (const uint4 *)fb.src0[blockIdx.y] : (const uint4 *)x0;
	const uint4 *x1 = (const uint4 *)fb.x1[blockIdx.y];
	const uint64_t n = fl.n[blockIdx.y];
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n; i += U * stride) {
		uint4 a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			a[u] = s0[i + u * stride];
			b[u] = x1[i + u * stride];
		}
#pragma unroll
		for (int u = 0; u < U; u++)
			x0[i + u * stride] = xor4(a[u], ctable_mul(tab, xor4(a[u], b[u])));
	}
	for (; i < n; i += stride) {
		uint4 a = s0[i], b = x1[i];
		x0[i] = xor4(a, ctable_mul(tab, xor4(a, b)));
	}
}

hipError_t launch_extrapolate_line_ragged(hipStream_t s, int n_cu, const fold_batch &b, const fold_lengths &fl, uint32_t count, f128 z)
{
	if (count == 0 || count > (uint32_t)kFoldBatchMax) return count ? hipErrorNotSupported : hipSuccess;
	uint64_t n_max = 0;
	for (uint32_t i = 0; i < count; i++) n_max = fl.n[i] > n_max ? fl.n[i] : n_max;
	if (n_max == 0) return hipSuccess;
	if (n_max >= (1u << 16)) {
		unsigned g = grid_for(n_max, 256 * 2, n_cu, 8);
		g = (g + count - 1) / count;
		if (g < 1) g = 1;
		hipLaunchKernelGGL((k_extrapolate_line_ragged<2>), dim3(g, count), dim3(256), 0, s, b, fl, z);
	} else {
		unsigned g = grid_for(n_max, 256, n_cu, 8);
		hipLaunchKernelGGL((k_extrapolate_line_ragged<1>), dim3(g, count), dim3(256), 0, s, b, fl, z);
	}
	return hipGetLastError();
}

hipError_t launch_fill(hipStream_t s, void *dst, uint64_t n, f128 v)
{
	if (n == 0) return hipSuccess;
	unsigned g = grid_for(n, 256, 256, 8);
	hipLaunchKernelGGL(k_fill, dim3(g), dim3(256), 0, s, (uint4 *)dst, n,
	                   uint4{(uint32_t)v.lo, (uint32_t)(v.lo >> 32), (uint32_t)v.hi, (uint32_t)(v.hi >> 32)});
	return hipGetLastError();
}

hipError_t launch_add_assign(hipStream_t s, void *dst, const void *src, uint64_t n)
{
	if (n == 0) return hipSuccess;
	unsigned g = grid_for(n, 256, 256, 8);
	hipLaunchKernelGGL(k_add_assign, dim3(g), dim3(256), 0, s, (uint4 *)dst, (const uint4 *)src, n);
	return hipGetLastError();
}

hipError_t launch_add(hipStream_t s, void *dst, const void *src1, const void *src2, uint64_t n)
{
	if (n == 0) return hipSuccess;
	unsigned g = grid_for(n, 256, 256, 8);
	hipLaunchKernelGGL(k_add, dim3(g), dim3(256), 0, s, (uint4 *)dst, (const uint4 *)src1, (const uint4 *)src2, n);
	return hipGetLastError();
}

// Launch geometry measured on MI355X (tools/fold_variants.hip, profiles/r01/fold_variants.txt):
// for HBM-resident sizes 2 blocks per CU, 2 elements in flight per lane and non-temporal accesses
// reach 5.8-5.9 TB/s; 8 blocks per CU only 4.6-4.8 TB/s.  Working sets that still fit the 256 MiB
// Infinity Cache (the n <= 24 rounds) are faster with the cached form and more blocks
// (70 us vs 87 us at r = 24), so the streaming form starts at 2^25 outputs (1.5 GiB of traffic).
hipError_t launch_extrapolate_line(hipStream_t s, int n_cu, void *evals_0, const void *evals_1, uint64_t n, f128 z)
{
	if (n == 0) return hipSuccess;
	uint4 *x0 = (uint4 *)evals_0;
	const uint4 *x1 = (const uint4 *)evals_1;
	if (n >= (1u << 25)) {
		hipLaunchKernelGGL((k_extrapolate_line<2, true>), dim3(2 * n_cu), dim3(256), 0, s, x0, x1, n, z);
	} else if (n >= (1u << 16)) {
		unsigned g = grid_for(n, 256 * 2, n_cu, 8);
		hipLaunchKernelGGL((k_extrapolate_line<2, false>), dim3(g), dim3(256), 0, s, x0, x1, n, z);
	} else {
		unsigned g = grid_for(n, 256, n_cu, 8);
		hipLaunchKernelGGL((k_extrapolate_line<1, false>), dim3(g), dim3(256), 0, s, x0, x1, n, z);
	}
	return hipGetLastError();
}

hipError_t launch_extrapolate_line_batch(hipStream_t s, int n_cu, const fold_batch &b, uint32_t count, uint64_t n, f128 z)
{
	if (n == 0 || count == 0) return hipSuccess;
	if (n * count >= (1u << 25)) {
		unsigned g = (2 * n_cu + count - 1) / count;
		hipLaunchKernelGGL((k_extrapolate_line_batch<2, true>), dim3(g, count), dim3(256), 0, s, b, n, z);
	} else if (n >= (1u << 16)) {
		unsigned g = grid_for(n, 256 * 2, n_cu, 8);
		g = (g + count - 1) / count;
		if (g < 1) g = 1;
		hipLaunchKernelGGL((k_extrapolate_line_batch<2, false>), dim3(g, count), dim3(256), 0, s, b, n, z);
	} else {
		unsigned g = grid_for(n, 256, n_cu, 8);
		hipLaunchKernelGGL((k_extrapolate_line_batch<1, false>), dim3(g, count), dim3(256), 0, s, b, n, z);
	}
	return hipGetLastError();
}

// The same batch fold for up to kFoldWideMax arrays per launch (3 KiB of kernel arguments): the deferred fold of a prover of the
// keccak width when it has to run by itself (abi_group.cpp launch_fold: a flush, or the arrays a group launch does not fold)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_extrapolate_line_wide(fold_batch_wide fb, uint64_t n, f128 z)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	uint4 *x0 = (uint4 *)fb.x0[blockIdx.y];
	const uint4 *s0 = fb.src0[blockIdx.y] ? (const uint4 *)fb.src0[blockIdx.y] : (const uint4 *)x0;
	const uint4 *x1 = (const uint4 *)fb.x1[blockIdx.y];
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n; i += U * stride) {
		uint4 a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			a[u] = ld16<NT>(&s0[i + u * stride]);
			b[u] = ld16<NT>(&x1[i + u * stride]);
		}
#pragma unroll
		for (int u = 0; u < U; u++)
			st16<NT>(&x0[i + u * stride], xor4(a[u], ctable_mul(tab, xor4(a[u], b[u]))));
	}
	for (; i < n; i += stride) {
		uint4 a = s0[i], b = x1[i];
		x0[i] = xor4(a, ctable_mul(tab, xor4(a, b)));
	}
}

hipError_t launch_extrapolate_line_wide(hipStream_t s, int n_cu, const fold_batch_wide &b, uint32_t count, uint64_t n, f128 z)
{
	if (n == 0 || count == 0) return hipSuccess;
	if (count > (uint32_t)kFoldWideMax) return hipErrorNotSupported;
	if (n * count >= (1u << 25)) {
		unsigned g = (2 * n_cu + count - 1) / count;
		hipLaunchKernelGGL((k_extrapolate_line_wide<2, true>), dim3(g, count), dim3(256), 0, s, b, n, z);
	} else if (n >= (1u << 16)) {
		unsigned g = grid_for(n, 256 * 2, n_cu, 8);
		g = (g + count - 1) / count;
		if (g < 1) g = 1;
		hipLaunchKernelGGL((k_extrapolate_line_wide<2, false>), dim3(g, count), dim3(256), 0, s, b, n, z);
	} else {
		unsigned g = grid_for(n, 256, n_cu, 8);
		g = (g + count - 1) / count; // (a hundred arrays of a few thousand elements: a workgroup or two each)
		if (g < 1) g = 1;
		hipLaunchKernelGGL((k_extrapolate_line_wide<1, false>), dim3(g, count), dim3(256), 0, s, b, n, z);
	}
	return hipGetLastError();
}

// A tiny fold batch (count * n <= 64 elements) whose results the host is about to read: fold in
// place and mirror every folded element into the pinned host mailbox (mail[arr * n + i]), then the
// sequence word.  The host-side copy_d2h calls that follow are served from the mailbox.
struct fold_publish_args {
	void *x0[kFoldBatchMax];
	const void *src0[kFoldBatchMax];
	const void *x1[kFoldBatchMax];
};
__global__ __launch_bounds__(256) void k_fold_publish(fold_publish_args fb, uint32_t count, uint32_t n, f128 z, f128 *mail, uint64_t seq,
                                                      uint32_t scale_mask, f128 hi_scale)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	const unsigned i = threadIdx.x;
	if (i < count * n) {
		const unsigned arr = i / n, j = i - arr * n;
		const uint4 a = ((const uint4 *)fb.src0[arr])[j], b = ((const uint4 *)fb.x1[arr])[j];
		uint4 f = xor4(a, ctable_mul(tab, xor4(a, b)));
		if (((scale_mask >> arr) & 1) && 2 * j >= n) f = to_u4(mul_slow(to_f128(f), hi_scale)); // (a handful of elements)
		((uint4 *)fb.x0[arr])[j] = f;
		const f128 v = to_f128(f);
		__hip_atomic_store(&mail[i].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&mail[i].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	// count * n <= 64: all value stores were issued by wave 0, whose release drains them first
	if (i == 0)
		__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_fold_publish(hipStream_t s, void *const *x0, const void *const *src0, const void *const *x1, uint32_t count, uint32_t n,
                               f128 z, f128 *d_mail, uint64_t seq, uint32_t scale_mask, f128 hi_scale)
{
	if (count == 0 || n == 0 || count > (uint32_t)kFoldBatchMax || (uint64_t)count * n > 64) return hipErrorNotSupported;
	fold_publish_args fb{};
	for (uint32_t i = 0; i < count; i++) {
		fb.x0[i] = x0[i];
		fb.src0[i] = src0[i];
		fb.x1[i] = x1[i];
	}
	hipLaunchKernelGGL(k_fold_publish, dim3(1), dim3(256), 0, s, fb, count, n, z, d_mail, seq, scale_mask, hi_scale);
	return hipGetLastError();
}

// The last TWO folds of a sumcheck in one launch (a round between them was answered from precomputed sums, abi_kernels.cpp):
// X (4 n_out elements: src0 | x1 are its halves) -> X' = X0 + z1 (X0 + X1) -> Y = X'0 + z2 (X'0 + X'1), n_out elements.
// Memory ends up exactly as after two separate in-place folds: Y in out[0, n_out), the upper half of X' in
// out[n_out, 2 n_out).  The two nibble tables are built side by side by the two halves of the workgroup.
__global__ __launch_bounds__(256) void k_fold2_publish(fold_publish_args fb, uint32_t count, uint32_t n_out, f128 z1, f128 z2, f128 *mail, uint64_t seq)
{
	__shared__ ctable_smem tab[2];
	ctable_build_group(tab[threadIdx.x >> 7], (threadIdx.x >> 7) ? z2 : z1, threadIdx.x & 127, 128);
	const unsigned i = threadIdx.x;
	if (i < count * n_out) {
		const unsigned arr = i / n_out, j = i - arr * n_out;
		const uint4 *lo = (const uint4 *)fb.src0[arr], *hi = (const uint4 *)fb.x1[arr];
		const uint4 a0 = lo[j], b0 = hi[j], a1 = lo[j + n_out], b1 = hi[j + n_out];
		const uint4 u = xor4(a0, ctable_mul(tab[0], xor4(a0, b0)));
		const uint4 v = xor4(a1, ctable_mul(tab[0], xor4(a1, b1)));
		const uint4 f = xor4(u, ctable_mul(tab[1], xor4(u, v)));
		((uint4 *)fb.x0[arr])[j] = f;
		((uint4 *)fb.x0[arr])[j + n_out] = v;
		const f128 r = to_f128(f);
		__hip_atomic_store(&mail[i].lo, r.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&mail[i].hi, r.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (i == 0)
		__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_fold2_publish(hipStream_t s, void *const *out, const void *const *src0, const void *const *x1, uint32_t count, uint32_t n_out,
                                f128 z1, f128 z2, f128 *d_mail, uint64_t seq)
{
	if (count == 0 || n_out == 0 || count > (uint32_t)kFoldBatchMax || (uint64_t)count * n_out > 64) return hipErrorNotSupported;
	fold_publish_args fb{};
	for (uint32_t i = 0; i < count; i++) {
		fb.x0[i] = out[i];
		fb.src0[i] = src0[i];
		fb.x1[i] = x1[i];
	}
	hipLaunchKernelGGL(k_fold2_publish, dim3(1), dim3(256), 0, s, fb, count, n_out, z1, z2, d_mail, seq);
	return hipGetLastError();
}

// The device catching up with a host tail (abi_kernels.cpp): the host has folded its copy of the two arrays -- in the power basis of
// hostmul_clmul.cpp -- and left it in pinned memory; out[j][i] = PhiInv(staging[j * n0 + i]) puts into the caller's buffers exactly what
// the folds it asked for would have (field arithmetic is exact: the same element whichever basis the products were taken in).  One
// element per thread, the inverse basis change as one nibble-table product; nobody waits for it.
__global__ __launch_bounds__(256) void k_tail_writeback(tail_writeback_args a, const uint4 *__restrict__ staging, const uint4 *__restrict__ phi_inv)
{
	__shared__ uint4 T[512];
	const unsigned tid = threadIdx.x, g = blockIdx.x * 256 + tid;
	const bool act = g < 2 * a.n0;
	const unsigned arr = g >= a.n0 ? 1 : 0, i = g - arr * a.n0;
	uint4 v{0, 0, 0, 0};
	if (act) v = staging[g];
	T[tid] = phi_inv[tid];
	T[tid + 256] = phi_inv[tid + 256];
	__syncthreads();
	if (act) ((uint4 *)a.out[arr])[i] = ctable_mul(*reinterpret_cast<const ctable_smem *>(T), v);
}

hipError_t launch_tail_writeback(hipStream_t s, const tail_writeback_args &a, const void *d_staging, const void *d_phi_inv)
{
	if (a.n0 == 0 || a.n0 > kHtMaxM / 2) return hipErrorNotSupported;
	hipLaunchKernelGGL(k_tail_writeback, dim3((2 * a.n0 + 255) / 256), dim3(256), 0, s, a, (const uint4 *)d_staging, (const uint4 *)d_phi_inv);
	return hipGetLastError();
}

// x[i] *= c in place (the stand-alone form of the upper-half scaling of bn_extrapolate_line_batch_scaled; the fused
// fold + evaluation kernels do it on the folded registers)
__global__ __launch_bounds__(256) void k_scale(uint4 *__restrict__ x, uint64_t n, f128 c)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, c);
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		x[i] = ctable_mul(tab, x[i]);
}

hipError_t launch_scale(hipStream_t s, int n_cu, void *x, uint64_t n, f128 c)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 256, n_cu, 8)), dim3(256), 0, s, (uint4 *)x, n, c);
	return hipGetLastError();
}

// d_out[0] ^= XOR_i x[i]: the sum of a lone row (a degree-1 term of a compiled circuit's sum, abi_circuit.cpp) at streaming
// speed -- the product-sum kernels would multiply it by an all-ones table
__global__ __launch_bounds__(256) void k_xor_sum(const uint4 *__restrict__ x, uint64_t n, f128 *out)
{
	__shared__ uint4 red[4];
	uint4 acc{0, 0, 0, 0};
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		acc = xor4(acc, x[i]);
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) {
		acc.x ^= __shfl_xor(acc.x, m, 64);
		acc.y ^= __shfl_xor(acc.y, m, 64);
		acc.z ^= __shfl_xor(acc.z, m, 64);
		acc.w ^= __shfl_xor(acc.w, m, 64);
	}
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x < 2) {
		const uint4 a = red[0], b = red[1], c = red[2], d = red[3];
		const uint64_t v = threadIdx.x == 0 ? ((uint64_t)(a.x ^ b.x ^ c.x ^ d.x) | ((uint64_t)(a.y ^ b.y ^ c.y ^ d.y) << 32))
		                                    : ((uint64_t)(a.z ^ b.z ^ c.z ^ d.z) | ((uint64_t)(a.w ^ b.w ^ c.w ^ d.w) << 32));
		if (v) atomicXor(reinterpret_cast<unsigned long long *>(out) + threadIdx.x, (unsigned long long)v);
	}
}

hipError_t launch_xor_sum(hipStream_t s, int n_cu, const void *x, uint64_t n, f128 *d_out)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_xor_sum, dim3(grid_for(n, 256 * 8, n_cu, 4)), dim3(256), 0, s, (const uint4 *)x, n, d_out);
	return hipGetLastError();
}

// out[i] = c * x[i] (a Mul(const, x) step of a compiled circuit, abi_circuit.cpp; out may be x)
__global__ __launch_bounds__(256) void k_scale_to(uint4 *out, const uint4 *x, uint64_t n, f128 c)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, c);
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		out[i] = ctable_mul(tab, x[i]);
}

hipError_t launch_scale_to(hipStream_t s, int n_cu, void *out, const void *x, uint64_t n, f128 c)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(k_scale_to, dim3(grid_for(n, 256, n_cu, 8)), dim3(256), 0, s, (uint4 *)out, (const uint4 *)x, n, c);
	return hipGetLastError();
}

// ---- tensor_expand, fused ---------------------------------------------------------------------
// (1) head: while the expansion fits in LDS (<= 2^12 elements) ALL passes run inside one workgroup.
struct expand_coords {
	f128 r[16];
};
constexpr int kHeadLog = 12;
// Up to four coordinates per step: the four quarters of the workgroup build one nibble table each at
// the same time (the table build, not the arithmetic, dominates these tiny passes); the passes
// themselves stay one multiplication per thread, a barrier apart.
__global__ __launch_bounds__(1024) void k_tensor_expand_head(uint4 *__restrict__ x, uint32_t log_n, uint32_t n_pass, expand_coords rc)
{
	extern __shared__ __attribute__((aligned(16))) uint4 head_buf[]; // 2^(log_n + n_pass) elements
	__shared__ ctable_smem tabs[4];
	const unsigned tid = threadIdx.x, grp = tid >> 8, ltid = tid & 255;
	for (unsigned h = tid; h < (1u << log_n); h += 1024)
		head_buf[h] = x[h];
	for (uint32_t i = 0; i < n_pass; i += 4) {
		const uint32_t P = (n_pass - i) < 4 ? (n_pass - i) : 4;
		// (the first barrier inside also orders the previous pass's / the load's LDS writes)
		ctable_build_group(tabs[grp], rc.r[(i + grp) & 15], grp < P ? ltid : 256u, 256u);
		for (uint32_t q = 0; q < P; q++) {
			const unsigned half = 1u << (log_n + i + q);
			for (unsigned h = tid; h < half; h += 1024) {
				const uint4 v = head_buf[h];
				const uint4 p = ctable_mul(tabs[q], v);
				head_buf[h] = xor4(v, p);
				head_buf[half + h] = p;
			}
			__syncthreads();
		}
	}
	for (unsigned h = tid; h < (1u << (log_n + n_pass)); h += 1024)
		x[h] = head_buf[h];
}

// (2) body: P consecutive passes in one streaming kernel.  Input element h (< half) ends up in 2^P
// places: after pass q the value w splits into w - w*r_q (same place) and w*r_q (place + half * 2^q).
// Reads 16 B, writes 2^P * 16 B per input element instead of (2^P - 1) reads + 2 (2^P - 1) writes.
template <int P>
struct expand_tabs {
	ctable_smem t[P];
};
template <int P>
__global__ __launch_bounds__(256) void k_tensor_expand_multi(uint4 *__restrict__ x, uint64_t half, expand_coords rc)
{
	__shared__ expand_tabs<P> tabs;
#pragma unroll
	for (int q = 0; q < P; q++)
		ctable_build(tabs.t[q], rc.r[q]);
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	for (uint64_t h = (uint64_t)blockIdx.x * 256 + threadIdx.x; h < half; h += stride) {
		uint4 w[1 << P];
		w[0] = x[h];
#pragma unroll
		for (int q = 0; q < P; q++) {
#pragma unroll
			for (int e = 0; e < (1 << q); e++) {
				const uint4 p = ctable_mul(tabs.t[q], w[e]);
				w[e] = xor4(w[e], p);
				w[e + (1 << q)] = p;
			}
		}
#pragma unroll
		for (int e = 0; e < (1 << P); e++)
			x[h + (uint64_t)e * half] = w[e];
	}
}

// all k passes of a tensor_expand (layer.rs:269-296)
hipError_t launch_tensor_expand(hipStream_t s, int n_cu, void *data, uint32_t log_n, const f128 *coords, uint32_t k)
{
	uint32_t i = 0;
	if (log_n < (uint32_t)kHeadLog && k > 0) {
		uint32_t n_pass = (uint32_t)kHeadLog - log_n;
		if (n_pass > k) n_pass = k;
		if (n_pass > 16) n_pass = 16;
		expand_coords rc{};
		for (uint32_t q = 0; q < n_pass; q++) rc.r[q] = coords[q];
		const size_t lds = ((size_t)16 << (log_n + n_pass));
		{
			const hipError_t e = func_lds_limit(reinterpret_cast<const void *>(&k_tensor_expand_head), (int)((size_t)16 << kHeadLog));
			if (e != hipSuccess) return e;
		}
		hipLaunchKernelGGL(k_tensor_expand_head, dim3(1), dim3(1024), lds, s, (uint4 *)data, log_n, n_pass, rc);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		i = n_pass;
	}
	while (i < k) {
		const uint32_t left = k - i;
		// groups of 3 from here; a remainder of 1 or 2 goes first so the largest passes are 3-fused
		const uint32_t P = (left % 3) ? (left % 3) : 3;
		const uint64_t half = (uint64_t)1 << (log_n + i);
		expand_coords rc{};
		for (uint32_t q = 0; q < P; q++) rc.r[q] = coords[i + q];
		const unsigned g = grid_for(half, 256, n_cu, P == 3 ? 4 : 8);
		if (P == 1)
			hipLaunchKernelGGL((k_tensor_expand_multi<1>), dim3(g), dim3(256), 0, s, (uint4 *)data, half, rc);
		else if (P == 2)
			hipLaunchKernelGGL((k_tensor_expand_multi<2>), dim3(g), dim3(256), 0, s, (uint4 *)data, half, rc);
		else
			hipLaunchKernelGGL((k_tensor_expand_multi<3>), dim3(g), dim3(256), 0, s, (uint4 *)data, half, rc);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
		i += P;
	}
	return hipSuccess;
}

hipError_t func_lds_limit(const void *fn, int bytes)
{
	struct entry {
		const void *fn;
		int dev, bytes;
	};
	static std::mutex mu;
	static std::vector<entry> done;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) {
		(void)hipGetLastError();
		dev = -1;
	}
	std::lock_guard<std::mutex> lk(mu);
	for (auto &d : done)
		if (d.fn == fn && d.dev == dev) {
			if (d.bytes >= bytes) return hipSuccess;
			const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
			if (e == hipSuccess) d.bytes = bytes;
			return e;
		}
	const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
	if (e == hipSuccess && dev >= 0) done.push_back(entry{fn, dev, bytes});
	return e;
}

} // namespace bn
