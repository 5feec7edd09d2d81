// binius_amd/csrc/kernels_ip32.hip -- inner_product of a BinaryField32b column with an F column
// (crates/compute/src/layer.rs:263, cpu/layer.rs:205-236 at tower level 5; FSub = B32 is the
// production small field, core/src/constraint_system/common.rs:22):   sum_i b[i] * a_sub[i].
//
// A B32 scalar acts on the four 32-bit limbs of an F element independently (F is a T_5-vector
// space with basis 1, X_5, X_6, X_5 X_6 = the limbs), so the inner product is FOUR independent
// GF(2^32) inner products  r_k = sum_i b_k[i] * s[i]  -- no Karatsuba across limbs, and everything
// is linear in the products, so each lane just accumulates 32 planes for the whole kernel.
//
// Wave = 16 groups of 4 lanes (lane & 3 = limb k).  A batch is 512 elements: row j of group g is
// element base + 16 j + g, so one wave-wide load instruction reads 256 contiguous bytes of b
// (16 elements x 4 words) resp. 64 contiguous bytes of s.  Per batch and lane: 32 + 32 four-byte
// loads, two 32x32 bit transposes, one bit-sliced GF(2^32) product (bitslice.hpp): ~1600 VALU per
// 512 elements = 3 VALU per element (the word-level bilinear walk needs ~250).
#include <hip/hip_runtime.h>

#include "bitslice.hpp"
#include "internal.hpp"

namespace bn {

namespace {
constexpr int kIpBatch = 512;
}

__global__ __launch_bounds__(256, 2) void k_ip32(const uint32_t *__restrict__ s, const uint32_t *__restrict__ b, uint64_t n, f128 *out)
{
	__shared__ uint32_t red[4][4];
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned k = lane & 3, g = lane >> 2;
	uint32_t acc[32];
#pragma unroll
	for (int p = 0; p < 32; p++)
		acc[p] = 0;
	const uint64_t n_batches = n / kIpBatch; // n is a multiple of 512 (checked by the launcher)
	const uint64_t wave_global = (uint64_t)blockIdx.x * 4 + wave;
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;
	const uint32_t *qb = b + 4 * g + k; // word k of element g
	const uint32_t *qs = s + g;
	uint32_t rb[32];
	auto load_b = [&](uint64_t bt) {
		const uint32_t *p = qb + bt * (kIpBatch * 4);
#pragma unroll
		for (int j = 0; j < 32; j++)
			rb[j] = p[64 * j]; // 16 elements = 64 words per row
	};
	if (wave_global < n_batches)
		load_b(wave_global);
	for (uint64_t bt = wave_global; bt < n_batches; bt += n_waves) {
		uint32_t A[32], B[32], P[32];
		{
			const uint32_t *p = qs + bt * kIpBatch;
#pragma unroll
			for (int j = 0; j < 32; j++)
				B[j] = p[16 * j];
		}
#pragma unroll
		for (int j = 0; j < 32; j++)
			A[j] = rb[j];
		if (bt + n_waves < n_batches)
			load_b(bt + n_waves); // next batch's rows fly during the transposes and the product
		transpose32(A);
		transpose32(B);
		bs_mul<5>(A, B, P);
#pragma unroll
		for (int p = 0; p < 32; p++)
			acc[p] ^= P[p];
	}
	// limb k of this lane's partial sum: bit p = parity of plane p
	uint32_t v = 0;
#pragma unroll
	for (int p = 0; p < 32; p++)
		v |= (__popc(acc[p]) & 1u) << p;
	// XOR over the 16 groups of the wave (lanes with the same k), then over the 4 waves
#pragma unroll
	for (int m = 32; m >= 4; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	if (lane < 4) red[wave][lane] = v;
	__syncthreads();
	if (threadIdx.x < 4) {
		const uint32_t r = red[0][threadIdx.x] ^ red[1][threadIdx.x] ^ red[2][threadIdx.x] ^ red[3][threadIdx.x];
		if (r) atomicXor(reinterpret_cast<unsigned int *>(out) + threadIdx.x, r);
	}
}

// d_out[0] ^= sum_i b[i] * s[i]; n elements, n a multiple of 512
hipError_t launch_ip32(hipStream_t st, int n_cu, const void *s, const void *b, uint64_t n, f128 *d_out)
{
	if (n == 0 || n % kIpBatch) return hipErrorNotSupported;
	const uint64_t n_batches = n / kIpBatch;
	uint64_t blocks = (n_batches + 3) / 4;
	const uint64_t cap = (uint64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(k_ip32, dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t *)s, (const uint32_t *)b, n, d_out);
	return hipGetLastError();
}

} // namespace bn
