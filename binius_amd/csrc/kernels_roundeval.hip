// binius_amd/csrc/kernels_roundeval.hip -- sumcheck round evaluation: sums of products of
// GF(2^128) multilinears (the body of accumulate_kernels + sum_composition_evals,
// crates/compute/src/layer.rs:183, 552; caller crates/core/src/protocols/sumcheck/v3/
// bivariate_product.rs:303-408).
//
// This is the only place on the measured path with variable x variable products.  They are done
// bit-sliced (bitslice.hpp): every lane transposes 32 rows of raw 128-bit elements into 128 bit
// planes with four in-register 32x32 transposes, multiplies plane sets with the tower recursion
// on AND/XOR (v_bitop3_b32), and accumulates the product planes in registers.  Because field
// addition is XOR, the whole sum over the hypercube is carried as 128 accumulated planes per lane
// and collapsed to two field elements only once per workgroup (parity of popcounts, then a
// wave/LDS XOR tree, then 64-bit atomicXor -- exact and order-independent).
//
// A register holds two 16-element groups: the low half carries the operands of the evaluation at
// 1 (the hi halves of the multilinears) and the high half the operands of the evaluation at
// infinity (lo + hi), so one pass over the data yields both round evaluations and the `Local`
// scratch buffers of the reference kernel (KernelMemMap::Local, layer.rs:608-611) never exist in
// HBM: "lo + hi" is a single XOR on the transposed planes.
#include <hip/hip_runtime.h>

#include "bitslice.hpp"
#include "ctable.hpp"
#include "finalize.hpp"
#include "internal.hpp"

namespace bn {

// how the two 16-row groups of a packed operand are filled
enum : uint32_t {
	BS_PAIR_XOR = 0, // low = p[i], high = p[i] ^ q[i]
	BS_DUP = 1,      // low = high = p[i]
	BS_SPLIT = 2,    // low = p[i], high = p[i + split_off]
};

struct bs_var {
	const uint4 *p;
	const uint4 *q;
	uint32_t mode;
};

constexpr int kMaxProdVars = 4;

struct bs_job {
	bs_var v[kMaxProdVars];
	uint32_t k;         // number of factors, 1..kMaxProdVars
	uint64_t n;         // elements per 16-row group stream
	uint64_t split_off; // BS_SPLIT only
};

__device__ __forceinline__ uint4 ld_or_zero(const uint4 *p, uint64_t i, bool ok)
{
	return ok ? p[i] : uint4{0, 0, 0, 0};
}

// Load 16 (+16) rows for one factor and turn them into 128 packed planes.
// Element j of the batch is index base + j*64 (lane already folded into base) for coalescing.
__device__ __forceinline__ void bs_load_operand(const bs_var &v, uint64_t base, uint64_t n, uint64_t split_off,
                                                uint32_t (&pl)[128])
{
	uint32_t r0[32], r1[32], r2[32], r3[32];
#pragma unroll
	for (int j = 0; j < 16; j++) {
		const uint64_t i = base + (uint64_t)j * 64;
		const bool ok = i < n;
		uint4 x = ld_or_zero(v.p, i, ok);
		uint4 y{0, 0, 0, 0};
		if (v.mode == BS_PAIR_XOR)
			y = ld_or_zero(v.q, i, ok);
		else if (v.mode == BS_SPLIT)
			y = ld_or_zero(v.p, i + split_off, ok);
		r0[j] = x.x;
		r1[j] = x.y;
		r2[j] = x.z;
		r3[j] = x.w;
		r0[16 + j] = y.x;
		r1[16 + j] = y.y;
		r2[16 + j] = y.z;
		r3[16 + j] = y.w;
	}
	transpose32(r0);
	transpose32(r1);
	transpose32(r2);
	transpose32(r3);
	if (v.mode == BS_PAIR_XOR) {
#pragma unroll
		for (int p = 0; p < 32; p++) {
			pl[p] = r0[p] ^ (r0[p] << 16);
			pl[32 + p] = r1[p] ^ (r1[p] << 16);
			pl[64 + p] = r2[p] ^ (r2[p] << 16);
			pl[96 + p] = r3[p] ^ (r3[p] << 16);
		}
	} else if (v.mode == BS_DUP) {
#pragma unroll
		for (int p = 0; p < 32; p++) {
			pl[p] = r0[p] | (r0[p] << 16);
			pl[32 + p] = r1[p] | (r1[p] << 16);
			pl[64 + p] = r2[p] | (r2[p] << 16);
			pl[96 + p] = r3[p] | (r3[p] << 16);
		}
	} else {
#pragma unroll
		for (int p = 0; p < 32; p++) {
			pl[p] = r0[p];
			pl[32 + p] = r1[p];
			pl[64 + p] = r2[p];
			pl[96 + p] = r3[p];
		}
	}
}

__device__ __forceinline__ uint32_t wave_xor(uint32_t v)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	return v;
}

// Collapse the per-lane accumulated planes to (S_low, S_high) and XOR them into out[0], out[1].
__device__ __forceinline__ void bs_finish(const uint32_t (&acc)[128], f128 *out)
{
	__shared__ uint32_t red[4][8];
	uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // w[0..3] = S_low words, w[4..7] = S_high words
#pragma unroll
	for (int p = 0; p < 128; p++) {
		uint32_t lo = __popc(acc[p] & 0xFFFFu) & 1u;
		uint32_t hi = __popc(acc[p] >> 16) & 1u;
		w[p >> 5] |= lo << (p & 31);
		w[4 + (p >> 5)] |= hi << (p & 31);
	}
#pragma unroll
	for (int t = 0; t < 8; t++)
		w[t] = wave_xor(w[t]);
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) {
#pragma unroll
		for (int t = 0; t < 8; t++)
			red[wave][t] = w[t];
	}
	__syncthreads();
	if (threadIdx.x < 4) {
		// thread t handles 64-bit word t of the 256-bit (S_low, S_high) pair
		const unsigned t = threadIdx.x;
		uint64_t v = 0;
		for (unsigned wv = 0; wv < (blockDim.x >> 6); wv++)
			v ^= (uint64_t)red[wv][2 * t] | ((uint64_t)red[wv][2 * t + 1] << 32);
		if (v)
			atomicXor(reinterpret_cast<unsigned long long *>(out) + t, (unsigned long long)v);
	}
}

// out[0] ^= sum over group-0 stream of prod_j v_j ; out[1] ^= same over the group-1 stream.
__global__ __launch_bounds__(256, 1) void k_bs_prodsum(bs_job job, f128 *out)
{
	uint32_t acc[128];
#pragma unroll
	for (int p = 0; p < 128; p++)
		acc[p] = 0;
	const unsigned lane = threadIdx.x & 63;
	const uint64_t wave_global = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const uint64_t n_waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	const uint64_t n_batches = (job.n + 1023) >> 10;
	for (uint64_t b = wave_global; b < n_batches; b += n_waves) {
		const uint64_t base = (b << 10) + lane;
		uint32_t P[128];
		bs_load_operand(job.v[0], base, job.n, job.split_off, P);
		for (uint32_t j = 1; j < job.k; j++) {
			uint32_t B[128], T[128];
			bs_load_operand(job.v[j], base, job.n, job.split_off, B);
			bs_mul<7>(P, B, T);
#pragma unroll
			for (int p = 0; p < 128; p++)
				P[p] = T[p];
		}
#pragma unroll
		for (int p = 0; p < 128; p++)
			acc[p] ^= P[p];
	}
	bs_finish(acc, out);
}

static unsigned prodsum_grid(uint64_t n, int n_cu)
{
	uint64_t n_batches = (n + 1023) >> 10;
	uint64_t blocks = (n_batches + 3) / 4;
	if (blocks < 1) blocks = 1;
	uint64_t cap = (uint64_t)n_cu; // one 256-thread block per CU: the kernel runs 1 wave per SIMD
	return (unsigned)(blocks < cap ? blocks : cap);
}

hipError_t launch_roundeval_product(hipStream_t s, int n_cu, const void *const *hi, const void *const *lo, uint32_t k,
                                    uint64_t n, f128 *d_out, const fin_fuse *fuse)
{
	if (k == 2 && lo[0] && lo[1]) {
		if (mfma_applies(n_cu, n)) return launch_roundeval_mfma_pair(s, n_cu, hi[0], lo[0], hi[1], lo[1], n, d_out, fuse);
		return launch_roundeval9_pair(s, n_cu, hi[0], lo[0], hi[1], lo[1], n, d_out, fuse);
	}
	if (k == 3) {
		// a * b * eq (MLE-check): exactly one factor is the same at both evaluation points
		int same = -1, n_same = 0;
		for (int j = 0; j < 3; j++)
			if (!lo[j]) {
				same = j;
				n_same++;
			}
		if (n_same == 1) {
			const int x = (same + 1) % 3, y = (same + 2) % 3;
			return launch_roundeval9_eq(s, n_cu, hi[x], lo[x], hi[y], lo[y], hi[same], n, d_out, fuse);
		}
	}
	if (fuse) return hipErrorNotSupported; // caller falls back to the stand-alone finalize kernel
	bs_job job{};
	job.k = k;
	job.n = n;
	job.split_off = 0;
	for (uint32_t j = 0; j < k; j++) {
		job.v[j].p = (const uint4 *)hi[j];
		job.v[j].q = (const uint4 *)lo[j];
		job.v[j].mode = lo[j] ? BS_PAIR_XOR : BS_DUP;
	}
	hipLaunchKernelGGL(k_bs_prodsum, dim3(prodsum_grid(n, n_cu)), dim3(256), 0, s, job, d_out);
	return hipGetLastError();
}

// d_out[0] ^= sum_i prod_j rows[j][i]   (d_out[1] is used as a second partial; caller XORs both)
hipError_t launch_sum_product(hipStream_t s, int n_cu, const void *const *rows, uint32_t n_rows, uint64_t row_len,
                              f128 *d_out)
{
	if (n_rows == 2 && row_len >= 2 && (row_len & 1) == 0) {
		if (mfma_applies(n_cu, row_len / 2)) return launch_roundeval_mfma_split(s, n_cu, rows[0], rows[1], row_len / 2, row_len / 2, d_out);
		return launch_roundeval9_split(s, n_cu, rows[0], rows[1], row_len / 2, row_len / 2, d_out);
	}
	bs_job job{};
	job.k = n_rows;
	if (row_len >= 2 && (row_len & 1) == 0) {
		job.n = row_len / 2; // two half-range streams, both halves of every register busy
		job.split_off = row_len / 2;
		for (uint32_t j = 0; j < n_rows; j++) {
			job.v[j].p = (const uint4 *)rows[j];
			job.v[j].q = nullptr;
			job.v[j].mode = BS_SPLIT;
		}
	} else {
		// odd length (CpuLayer takes any SlicesBatch row_len, cpu/layer.rs:168-249) or a single element:
		// everything in group 0 (group 1 = zero rows via PAIR_XOR with q = p)
		job.n = row_len;
		job.split_off = 0;
		for (uint32_t j = 0; j < n_rows; j++) {
			job.v[j].p = (const uint4 *)rows[j];
			job.v[j].q = (const uint4 *)rows[j];
			job.v[j].mode = BS_PAIR_XOR;
		}
	}
	hipLaunchKernelGGL(k_bs_prodsum, dim3(prodsum_grid(job.n, n_cu)), dim3(256), 0, s, job, d_out);
	return hipGetLastError();
}

// ---- generic circuit interpreter (slow path: arbitrary ArithCircuit, conformance only) ----------
constexpr int kMaxSteps = 64;

__device__ __forceinline__ f128 circuit_eval_dev(const bn_step *steps, uint32_t n_steps, const uint4 *const *rows, uint64_t i)
{
	f128 ev[kMaxSteps];
	for (uint32_t s = 0; s < n_steps; s++) {
		const bn_step st = steps[s];
		f128 r;
		switch (st.kind) {
		case BN_STEP_ADD: r = ev[st.a] ^ ev[st.b]; break;
		case BN_STEP_MUL: r = mul_slow(ev[st.a], ev[st.b]); break;
		case BN_STEP_POW: r = pow_slow(ev[st.a], st.b); break;
		case BN_STEP_CONST: r = f128{st.cst.lo, st.cst.hi}; break;
		default: r = to_f128(rows[st.a][i]); break;
		}
		ev[s] = r;
	}
	return n_steps ? ev[n_steps - 1] : f128_zero();
}

__global__ __launch_bounds__(256) void k_sum_composition_generic(const uint4 *const *rows, uint64_t row_len,
                                                                 const bn_step *steps, uint32_t n_steps, f128 *out)
{
	__shared__ uint64_t red[4][2];
	f128 acc = f128_zero();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < row_len; i += (uint64_t)gridDim.x * 256)
		acc ^= circuit_eval_dev(steps, n_steps, rows, i);
	uint32_t w[4] = {(uint32_t)acc.lo, (uint32_t)(acc.lo >> 32), (uint32_t)acc.hi, (uint32_t)(acc.hi >> 32)};
#pragma unroll
	for (int t = 0; t < 4; t++)
		w[t] = wave_xor(w[t]);
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) {
		red[wave][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
		red[wave][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
	}
	__syncthreads();
	if (threadIdx.x < 2) {
		uint64_t v = red[0][threadIdx.x] ^ red[1][threadIdx.x] ^ red[2][threadIdx.x] ^ red[3][threadIdx.x];
		if (v)
			atomicXor(reinterpret_cast<unsigned long long *>(out) + threadIdx.x, (unsigned long long)v);
	}
}

hipError_t launch_sum_composition_generic(hipStream_t s, int n_cu, const void *const *d_rows_dev, uint32_t n_rows,
                                          uint64_t row_len, const bn_step *d_steps, uint32_t n_steps, f128 *d_out)
{
	(void)n_rows;
	if (row_len == 0) return hipSuccess;
	uint64_t want = (row_len + 255) / 256;
	unsigned g = (unsigned)(want < (uint64_t)n_cu * 4 ? want : (uint64_t)n_cu * 4);
	hipLaunchKernelGGL(k_sum_composition_generic, dim3(g), dim3(256), 0, s, (const uint4 *const *)d_rows_dev, row_len,
	                   d_steps, n_steps, d_out);
	return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_compute_composite_generic(const uint4 *const *rows, uint64_t row_len, uint4 *out,
                                                                   const bn_step *steps, uint32_t n_steps)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < row_len; i += (uint64_t)gridDim.x * 256)
		out[i] = to_u4(circuit_eval_dev(steps, n_steps, rows, i));
}

hipError_t launch_compute_composite_generic(hipStream_t s, const void *const *d_rows_dev, uint32_t n_rows,
                                            uint64_t row_len, void *out, const bn_step *d_steps, uint32_t n_steps)
{
	(void)n_rows;
	if (row_len == 0) return hipSuccess;
	uint64_t want = (row_len + 255) / 256;
	unsigned g = (unsigned)(want < 4096 ? want : 4096);
	hipLaunchKernelGGL(k_compute_composite_generic, dim3(g), dim3(256), 0, s, (const uint4 *const *)d_rows_dev, row_len,
	                   (uint4 *)out, d_steps, n_steps);
	return hipGetLastError();
}

// ---- finalize: value[v] = init ^ XOR_t coeff_t * S[slot_t]; rets gathered ---------------------
__global__ __launch_bounds__(128) void k_finalize(fin_args a, f128 *S, f128 *rets, f128 *mail, fin_peer peer)
{
	finalize_body(a, S, rets, mail, a.seq, nullptr, &peer);
}

hipError_t launch_finalize(hipStream_t s, const fin_args &args, f128 *d_S, f128 *d_rets, f128 *d_mail, const fin_peer *peer)
{
	fin_peer pr{};
	if (peer) pr = *peer;
	hipLaunchKernelGGL(k_finalize, dim3(1), dim3(128), 0, s, args, d_S, d_rets, d_mail, pr);
	return hipGetLastError();
}

// out[i] = XOR_g vals[g * g_stride + i * i_stride], straight into the host mailbox (the combine step of the
// sharded prover after the per-round all_gather -- RCCL has no XOR reduction --, g_stride = group_len, i_stride = 1; the raw
// slot pairs of the old HAL's round evaluations, g_stride = 1, i_stride = 2)
__global__ __launch_bounds__(64) void k_xor_publish(const f128 *vals, uint32_t n_groups, uint32_t group_len, uint32_t g_stride, uint32_t i_stride, f128 *rets,
                                                    f128 *mail, uint64_t seq)
{
	const unsigned i = threadIdx.x;
	if (i < group_len) {
		f128 v = f128_zero();
		for (uint32_t g = 0; g < n_groups; g++)
			v ^= vals[(size_t)g * g_stride + (size_t)i * i_stride];
		rets[i] = v;
		__hip_atomic_store(&mail[i].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(&mail[i].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	__threadfence_system();
	__syncthreads();
	if (i == 0)
		__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_xor_publish(hipStream_t s, const f128 *d_vals, uint32_t n_groups, uint32_t group_len, f128 *d_rets, f128 *d_mail,
                              uint64_t seq, uint32_t g_stride, uint32_t i_stride)
{
	hipLaunchKernelGGL(k_xor_publish, dim3(1), dim3(64), 0, s, d_vals, n_groups, group_len, g_stride ? g_stride : group_len, i_stride ? i_stride : 1u, d_rets,
	                   d_mail, seq);
	return hipGetLastError();
}

} // namespace bn
