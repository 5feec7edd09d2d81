// binius_amd/csrc/kernels_ntt.hip -- additive NTT (LCH14 / DP24) forward and inverse.
//
// Semantics: crates/ntt/src/additive_ntt.rs:102,128; scalar definition
// crates/ntt/src/tests/reference.rs:68-160.  Layer i pairs (j<<(i+1)|k, ..|1<<i) with twiddle
// t = W_i(coset<<(log_y-1-i) | j), forward: u += v*t; v += u, inverse: v += u; u += v*t.
// Twiddles are never materialised in HBM: t is a GF(2)-linear function of the block index
// (crates/ntt/src/twiddle.rs:141-168), so the <= 63 basis values of a layer sit in LDS and a
// butterfly XOR-combines them.
//
// Round-1 implementation: one streaming pass per layer, element x twiddle via the SWAR bilinear
// walk of gf128.hpp.  (The LDS-blocked multi-layer / bit-sliced version is the planned upgrade.)
#include <hip/hip_runtime.h>

#include "ctable.hpp"
#include "internal.hpp"

namespace bn {

// a * b for a word of packed T_{>=K} limbs `a` and b in T_K (SWAR walk on 64 bits)
template <int K>
__device__ __forceinline__ uint64_t mul_walk64(uint64_t a, uint64_t bw)
{
	if constexpr (K == 0) {
		return a & (0 - (bw & 1));
	} else {
		constexpr int H = 1 << (K - 1);
		return mul_walk64<K - 1>(a, bw) ^ mul_walk64<K - 1>(mulx64<K - 1>(a), bw >> H);
	}
}

template <typename T>
struct elem_ops;

template <>
struct elem_ops<uint4> {
	template <int TW>
	static __device__ __forceinline__ uint4 mul_tw(uint4 v, uint64_t t)
	{
		return to_u4(mul_walk<TW>(to_f128(v), t));
	}
	static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) { return xor4(a, b); }
};

#define BN_SMALL_ELEM(T)                                                           \
	template <>                                                                    \
	struct elem_ops<T> {                                                           \
		template <int TW>                                                          \
		static __device__ __forceinline__ T mul_tw(T v, uint64_t t)                \
		{                                                                          \
			return (T)mul_walk64<TW>((uint64_t)v, t);                              \
		}                                                                          \
		static __device__ __forceinline__ T add(T a, T b) { return (T)(a ^ b); }   \
	};
BN_SMALL_ELEM(uint8_t)
BN_SMALL_ELEM(uint16_t)
BN_SMALL_ELEM(uint32_t)
BN_SMALL_ELEM(uint64_t)

template <typename T, int TW, bool INVERSE>
__global__ __launch_bounds__(256) void k_ntt_layer(T *data, const uint64_t *s_row, int n_bits, uint32_t log_x,
                                                   uint32_t log_y, uint32_t log_z, uint32_t layer, uint64_t coset)
{
	__shared__ uint64_t s_basis[64];
	if (threadIdx.x < 64)
		s_basis[threadIdx.x] = ((int)threadIdx.x < n_bits) ? s_row[threadIdx.x] : 0;
	__syncthreads();
	const uint64_t n_bfly = (uint64_t)1 << (log_x + log_y - 1 + log_z);
	const uint32_t i = layer;
	for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < n_bfly; q += (uint64_t)gridDim.x * 256) {
		const uint64_t x = q & (((uint64_t)1 << log_x) - 1);
		uint64_t rest = q >> log_x;
		const uint64_t k = rest & (((uint64_t)1 << i) - 1);
		rest >>= i;
		const uint64_t j = rest & (((uint64_t)1 << (log_y - 1 - i)) - 1);
		const uint64_t z = rest >> (log_y - 1 - i);
		const uint64_t tidx = (coset << (log_y - 1 - i)) | j;
		uint64_t t = 0;
		for (int b = 0; b < n_bits; b++)
			if ((tidx >> b) & 1)
				t ^= s_basis[b];
		const uint64_t idx0 = (j << (i + 1)) | k;
		const uint64_t idx1 = idx0 | ((uint64_t)1 << i);
		const uint64_t zoff = z << (log_x + log_y);
		const uint64_t p0 = x | (idx0 << log_x) | zoff;
		const uint64_t p1 = x | (idx1 << log_x) | zoff;
		T u = data[p0], v = data[p1];
		if (!INVERSE) {
			u = elem_ops<T>::add(u, elem_ops<T>::template mul_tw<TW>(v, t));
			v = elem_ops<T>::add(v, u);
		} else {
			v = elem_ops<T>::add(v, u);
			u = elem_ops<T>::add(u, elem_ops<T>::template mul_tw<TW>(v, t));
		}
		data[p0] = u;
		data[p1] = v;
	}
}

template <typename T, int TW>
static hipError_t run_layers(hipStream_t s, bool inverse, void *data, const uint64_t *d_s_evals, uint32_t log_domain,
                             uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset, uint32_t coset_bits,
                             uint32_t skip_rounds)
{
	const uint32_t base = log_domain - (log_y + coset_bits); // s_evals = &s_evals[base..] (reference.rs:89)
	const uint64_t n_bfly = (uint64_t)1 << (log_x + log_y - 1 + log_z);
	uint64_t want = (n_bfly + 255) / 256;
	unsigned g = (unsigned)(want < 4096 ? want : 4096);
	const int n_layers = (int)log_y - (int)skip_rounds;
	for (int step = 0; step < n_layers; step++) {
		const uint32_t i = inverse ? (uint32_t)step : (uint32_t)(n_layers - 1 - step);
		const uint32_t layer = base + i;
		const uint64_t *row = d_s_evals + (uint64_t)layer * BN_NTT_MAX_DIM;
		const int n_bits = (int)log_domain - 1 - (int)layer;
		if (inverse)
			hipLaunchKernelGGL((k_ntt_layer<T, TW, true>), dim3(g), dim3(256), 0, s, (T *)data, row, n_bits, log_x, log_y,
			                   log_z, i, coset);
		else
			hipLaunchKernelGGL((k_ntt_layer<T, TW, false>), dim3(g), dim3(256), 0, s, (T *)data, row, n_bits, log_x, log_y,
			                   log_z, i, coset);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

template <typename T>
static hipError_t run_tw(hipStream_t s, bool inverse, void *data, uint32_t tw_level, const uint64_t *d_s_evals,
                         uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset,
                         uint32_t coset_bits, uint32_t skip_rounds)
{
	switch (tw_level) {
	case 3: return run_layers<T, 3>(s, inverse, data, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 4: return run_layers<T, 4>(s, inverse, data, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 5: return run_layers<T, 5>(s, inverse, data, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 6: return run_layers<T, 6>(s, inverse, data, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	default: return hipErrorInvalidValue;
	}
}

hipError_t launch_ntt(hipStream_t s, bool inverse, void *data, uint32_t elem_level, uint32_t tw_level,
                      const uint64_t *d_s_evals, uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z,
                      uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds)
{
	switch (elem_level) {
	case 3: return run_tw<uint8_t>(s, inverse, data, tw_level, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 4: return run_tw<uint16_t>(s, inverse, data, tw_level, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 5: return run_tw<uint32_t>(s, inverse, data, tw_level, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 6: return run_tw<uint64_t>(s, inverse, data, tw_level, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 7: return run_tw<uint4>(s, inverse, data, tw_level, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	default: return hipErrorInvalidValue;
	}
}

} // namespace bn
