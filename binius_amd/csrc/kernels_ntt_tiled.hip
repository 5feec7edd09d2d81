// binius_amd/csrc/kernels_ntt_tiled.hip -- LDS-blocked additive NTT: several butterfly layers per
// pass over HBM, element x twiddle through an LDS-resident GF(2^8) product table.
//
// Semantics identical to kernels_ntt.hip / crates/ntt/src/tests/reference.rs:68-160.
//
// * A pass handles L consecutive layers [i_lo, i_hi].  A workgroup owns a tile of 2^T elements:
//   the L butterfly bits of the linear index plus `c` run bits (the lowest free index bits, for
//   coalescing); the tile lives in LDS for all L layers, so HBM is touched once per pass
//   (24 layers of 2^24 elements: 4 passes instead of 24).
// * Twiddles: t = W_i(coset << (log_y-1-i) | y >> (i+1)) is a GF(2)-linear function of the index
//   (crates/ntt/src/twiddle.rs:141-168).  Per layer the tile-varying low bits index a small LDS
//   table built by the workgroup; the tile-constant high bits contribute one XOR.
// * GF(2^32) x GF(2^32) (and 16-/8-bit twiddles) = Karatsuba over GF(2^8) with the 64 KiB product
//   table M[a][b] staged in LDS: 9 byte lookups + a few SWAR mul_alpha steps per 32-bit product
//   (the word-level SWAR walk of gf128.hpp costs ~900 VALU per product).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "ctable.hpp"
#include "internal.hpp"

namespace bn {

// ---- GF(2^8) product table (built once per context) -------------------------------------------
__global__ __launch_bounds__(256) void k_build_mul8(uint8_t *tab)
{
	const unsigned a = blockIdx.x, b = threadIdx.x;
	tab[(a << 8) | b] = (uint8_t)mul_walk<3>(f128{a, 0}, b).lo;
}

hipError_t launch_build_mul8(hipStream_t s, uint8_t *d_tab)
{
	hipLaunchKernelGGL(k_build_mul8, dim3(256), dim3(256), 0, s, d_tab);
	return hipGetLastError();
}

namespace {

// a = data byte, b = twiddle byte.  The table is symmetric; indexing the ROW by the twiddle byte puts
// the bank-selecting low address bits under the data byte, which differs across lanes -- lanes of a
// wave mostly share the twiddle, and with the roles swapped they all hit one bank (measured: 94 % of
// LDS cycles were bank conflicts).
__device__ __forceinline__ uint32_t m8(const uint8_t *M, uint32_t a, uint32_t b) { return M[(b << 8) | a]; }

// z * X_2 for a byte z (mul_alpha at level 3) is row 0x10 of the product table: one more lookup
// instead of ~25 SWAR VALU ops -- LDS has headroom here, VALU does not.
__device__ __forceinline__ uint32_t alpha8(const uint8_t *M, uint32_t z) { return M[(0x10u << 8) | z]; }

// 16-bit product: Karatsuba over GF(2^8); hi = z1' + lo + z2 * X_2
__device__ __forceinline__ uint32_t mul16_tab(const uint8_t *M, uint32_t a, uint32_t b)
{
	const uint32_t a0 = a & 0xFF, a1 = a >> 8, b0 = b & 0xFF, b1 = b >> 8;
	const uint32_t z0 = m8(M, a0, b0), z2 = m8(M, a1, b1), z1 = m8(M, a0 ^ a1, b0 ^ b1);
	const uint32_t lo = z0 ^ z2;
	const uint32_t hi = z1 ^ lo ^ alpha8(M, z2);
	return lo | (hi << 8);
}

// 32-bit product: Karatsuba over the 16-bit product; z2 * X_3 = (z2.hi, z2.lo + z2.hi * X_2)
__device__ __forceinline__ uint32_t mul32_tab(const uint8_t *M, uint32_t a, uint32_t b)
{
	const uint32_t a0 = a & 0xFFFF, a1 = a >> 16, b0 = b & 0xFFFF, b1 = b >> 16;
	const uint32_t z0 = mul16_tab(M, a0, b0), z2 = mul16_tab(M, a1, b1), z1 = mul16_tab(M, a0 ^ a1, b0 ^ b1);
	const uint32_t lo = z0 ^ z2;
	const uint32_t z2h = z2 >> 8, z2l = z2 & 0xFF;
	const uint32_t z2a = z2h | ((z2l ^ alpha8(M, z2h)) << 8);
	const uint32_t hi = z1 ^ lo ^ z2a;
	return lo | (hi << 16);
}

// one 32-bit word of an element times a twiddle of level TW (the twiddle multiplies every
// 2^TW-bit limb: crates/field/src/binary_field.rs:361-393)
template <int TW>
__device__ __forceinline__ uint32_t mulw_tab(const uint8_t *M, uint32_t w, uint32_t t)
{
	if constexpr (TW == 5) {
		return mul32_tab(M, w, t);
	} else if constexpr (TW == 4) {
		return mul16_tab(M, w & 0xFFFF, t & 0xFFFF) | (mul16_tab(M, w >> 16, t & 0xFFFF) << 16);
	} else {
		return m8(M, w & 0xFF, t) | (m8(M, (w >> 8) & 0xFF, t) << 8) | (m8(M, (w >> 16) & 0xFF, t) << 16) | (m8(M, w >> 24, t) << 24);
	}
}

template <typename T>
struct tile_ops;
template <>
struct tile_ops<uint32_t> {
	template <int TW>
	static __device__ __forceinline__ uint32_t mul(const uint8_t *M, uint32_t v, uint32_t t) { return mulw_tab<TW>(M, v, t); }
	static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return a ^ b; }
};
template <>
struct tile_ops<uint16_t> {
	template <int TW>
	static __device__ __forceinline__ uint16_t mul(const uint8_t *M, uint16_t v, uint32_t t)
	{
		if constexpr (TW == 4) return (uint16_t)mul16_tab(M, v, t);
		else return (uint16_t)(m8(M, v & 0xFF, t) | (m8(M, v >> 8, t) << 8));
	}
	static __device__ __forceinline__ uint16_t add(uint16_t a, uint16_t b) { return a ^ b; }
};
template <>
struct tile_ops<uint64_t> {
	template <int TW>
	static __device__ __forceinline__ uint64_t mul(const uint8_t *M, uint64_t v, uint32_t t)
	{
		return (uint64_t)mulw_tab<TW>(M, (uint32_t)v, t) | ((uint64_t)mulw_tab<TW>(M, (uint32_t)(v >> 32), t) << 32);
	}
	static __device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) { return a ^ b; }
};
template <>
struct tile_ops<uint4> {
	template <int TW>
	static __device__ __forceinline__ uint4 mul(const uint8_t *M, uint4 v, uint32_t t)
	{
		return uint4{mulw_tab<TW>(M, v.x, t), mulw_tab<TW>(M, v.y, t), mulw_tab<TW>(M, v.z, t), mulw_tab<TW>(M, v.w, t)};
	}
	static __device__ __forceinline__ uint4 add(uint4 a, uint4 b) { return xor4(a, b); }
};

struct ntt_pass {
	uint32_t log_x, log_y, log_z;
	uint32_t i_lo, i_hi;       // layers of this pass (y-bit indices), processed i_hi..i_lo forward, i_lo..i_hi inverse
	uint32_t r0, r1;           // run bits below / above the butterfly bits
	uint32_t base_layer;       // s_evals row of y-layer 0  (log_domain - (log_y + coset_bits))
	uint32_t log_domain;
	uint64_t coset;
	uint64_t n_tiles;
};

constexpr int kTwTabMaxBits = 11;

} // namespace

constexpr int kTiledThreads = 1024; // 16 waves per CU: the kernel alternates LDS lookups and VALU, it needs the TLP
constexpr int kMaxPassLayers = 6;

// dynamic LDS layout: [mul8 64 KiB][L twiddle tables of 2^kTwTabMaxBits u32][L x 64 basis rows u64][tile 2^T elements]
template <typename T, int TW, bool INVERSE>
__global__ __launch_bounds__(kTiledThreads) void k_ntt_tiled(T *data, const uint8_t *g_mul8, const uint64_t *s_evals, ntt_pass P)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint8_t *M = smem;
	uint32_t *twtabs = reinterpret_cast<uint32_t *>(smem + 65536);
	uint64_t *rowbufs = reinterpret_cast<uint64_t *>(smem + 65536 + kMaxPassLayers * (sizeof(uint32_t) << kTwTabMaxBits));
	T *tile = reinterpret_cast<T *>(smem + 65536 + kMaxPassLayers * (sizeof(uint32_t) << kTwTabMaxBits) + kMaxPassLayers * 64 * sizeof(uint64_t));

	const unsigned tid = threadIdx.x, nthr = blockDim.x;
	const uint32_t L = P.i_hi - P.i_lo + 1;
	const uint32_t Tbits = P.r0 + L + P.r1;
	const uint32_t B0 = P.log_x + P.i_lo, B1 = P.log_x + P.i_hi;
	const uint32_t g1 = B0 - P.r0;
	const uint32_t tile_n = 1u << Tbits;
	const uint32_t ybits_above = P.log_y - 1 - P.i_hi;           // y bits above the butterfly range
	const uint32_t r1y = P.r1 < ybits_above ? P.r1 : ybits_above; // run_hi bits that are y bits

	// ---- once per workgroup: product table, per-layer twiddle bases and tables (tile independent)
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(g_mul8);
		uint4 *dst = reinterpret_cast<uint4 *>(M);
		for (unsigned q = tid; q < 4096; q += nthr) dst[q] = src[q];
	}
	for (unsigned q = tid; q < L * 64; q += nthr) {
		const uint32_t li = q >> 6, b = q & 63;
		const uint32_t layer = P.base_layer + P.i_lo + li;
		const uint32_t n_bits = P.log_domain - 1 - layer;
		rowbufs[q] = (b < n_bits) ? s_evals[(uint64_t)layer * BN_NTT_MAX_DIM + b] : 0;
	}
	__syncthreads();
	for (uint32_t li = 0; li < L; li++) {
		const uint32_t i = P.i_lo + li;
		const uint32_t tab_bits = (P.i_hi - i) + r1y;
		const uint64_t *rb = rowbufs + li * 64;
		for (uint32_t e = tid; e < (1u << tab_bits); e += nthr) {
			uint64_t t = 0;
			for (uint32_t q = 0; q < tab_bits; q++)
				if ((e >> q) & 1) t ^= rb[q];
			twtabs[(li << kTwTabMaxBits) + e] = (uint32_t)t;
		}
	}

	for (uint64_t tv = blockIdx.x; tv < P.n_tiles; tv += gridDim.x) {
		const uint64_t g1v = tv & (((uint64_t)1 << g1) - 1), g2v = tv >> g1;
		const uint64_t lin_base = (g1v << P.r0) | (g2v << (B1 + 1 + P.r1));
		auto lin_of = [&](uint32_t u) -> uint64_t {
			const uint64_t a = u & ((1u << P.r0) - 1);
			const uint64_t t = (u >> P.r0) & ((1u << L) - 1);
			const uint64_t h = u >> (P.r0 + L);
			return lin_base | a | (t << B0) | (h << (B1 + 1));
		};
		// y bits of this tile above the tile-varying ones
		const uint64_t y_of_tile = ((g2v << (B1 + 1 + P.r1)) >> P.log_x) & ((((uint64_t)1) << P.log_y) - 1);
		const uint64_t yhigh = y_of_tile >> (P.i_hi + 1 + r1y);
		__syncthreads(); // previous tile fully stored; tables staged
		for (uint32_t u = tid; u < tile_n; u += nthr) tile[u] = data[lin_of(u)];

		for (uint32_t step = 0; step < L; step++) {
			const uint32_t i = INVERSE ? (P.i_lo + step) : (P.i_hi - step);
			const uint32_t li = i - P.i_lo;
			const uint32_t lb = P.r0 + li; // local bit of this layer
			const uint32_t layer = P.base_layer + i;
			const uint32_t n_bits = P.log_domain - 1 - layer; // bits of the twiddle index of this row
			const uint32_t nb_t = P.i_hi - i;                 // butterfly bits above i
			const uint32_t tab_bits = nb_t + r1y;
			const uint64_t *rb = rowbufs + li * 64;
			const uint32_t *twtab = twtabs + (li << kTwTabMaxBits);
			// tile-constant part of the twiddle: the remaining y bits and the coset (wave-uniform)
			const uint64_t jhigh = (P.coset << (P.log_y - 1 - i - tab_bits)) | yhigh; // j >> tab_bits
			uint64_t base_tw = 0;
			for (uint32_t q = 0; q + tab_bits < n_bits; q++)
				if ((jhigh >> q) & 1) base_tw ^= rb[tab_bits + q];
			__syncthreads(); // tile loaded / previous layer finished
			for (uint32_t bf = tid; bf < (tile_n >> 1); bf += nthr) {
				const uint32_t lowm = (1u << lb) - 1;
				const uint32_t u0 = ((bf & ~lowm) << 1) | (bf & lowm); // insert a 0 at local bit lb
				const uint32_t u1 = u0 | (1u << lb);
				const uint32_t tt = (u0 >> P.r0) & ((1u << L) - 1);
				const uint32_t hh = u0 >> (P.r0 + L);
				const uint32_t vidx = (tt >> (li + 1)) | ((hh & ((1u << r1y) - 1)) << nb_t);
				const uint32_t tw = (uint32_t)base_tw ^ twtab[vidx];
				T uu = tile[u0], vv = tile[u1];
				if (!INVERSE) {
					uu = tile_ops<T>::add(uu, tile_ops<T>::template mul<TW>(M, vv, tw));
					vv = tile_ops<T>::add(vv, uu);
				} else {
					vv = tile_ops<T>::add(vv, uu);
					uu = tile_ops<T>::add(uu, tile_ops<T>::template mul<TW>(M, vv, tw));
				}
				tile[u0] = uu;
				tile[u1] = vv;
			}
		}
		__syncthreads();
		for (uint32_t u = tid; u < tile_n; u += nthr) data[lin_of(u)] = tile[u];
	}
}

template <typename T, int TW>
static hipError_t run_tiled(hipStream_t s, int n_cu, bool inverse, void *data, const uint8_t *d_mul8, const uint64_t *d_s_evals,
                            uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset, uint32_t coset_bits,
                            uint32_t skip_rounds)
{
	const uint32_t n_layers = log_y - skip_rounds; // layers 0 .. n_layers-1 (forward: high to low)
	const uint32_t Ltot = log_x + log_y + log_z;
	const uint32_t Tmax = sizeof(T) >= 16 ? 10 : 11; // tile elements (2^T) : 16 KiB for 128-bit, 8 KiB for 32-bit
	const uint32_t Lmax = kMaxPassLayers;
	// split the layers into passes of <= Lmax, aligned from the top for the forward direction
	std::vector<std::pair<uint32_t, uint32_t>> passes; // (i_lo, i_hi)
	{
		int hi = (int)n_layers - 1;
		while (hi >= 0) {
			int lo = hi - (int)Lmax + 1;
			if (lo < 0) lo = 0;
			passes.push_back({(uint32_t)lo, (uint32_t)hi});
			hi = lo - 1;
		}
		if (inverse) std::reverse(passes.begin(), passes.end());
	}
	for (auto &pr : passes) {
		ntt_pass P{};
		P.log_x = log_x;
		P.log_y = log_y;
		P.log_z = log_z;
		P.i_lo = pr.first;
		P.i_hi = pr.second;
		const uint32_t L = P.i_hi - P.i_lo + 1;
		const uint32_t B0 = log_x + P.i_lo, B1 = log_x + P.i_hi;
		uint32_t c = Tmax > L ? Tmax - L : 0;
		if (c > Ltot - L) c = Ltot - L;
		P.r0 = c < B0 ? c : B0;
		P.r1 = c - P.r0;
		if (P.r1 > Ltot - B1 - 1) P.r1 = Ltot - B1 - 1;
		P.base_layer = log_domain - (log_y + coset_bits);
		P.log_domain = log_domain;
		P.coset = coset;
		const uint32_t Tbits = P.r0 + L + P.r1;
		P.n_tiles = (uint64_t)1 << (Ltot - Tbits);
		const size_t lds = 65536 + kMaxPassLayers * (sizeof(uint32_t) << kTwTabMaxBits) + kMaxPassLayers * 64 * sizeof(uint64_t) + (sizeof(T) << Tbits);
		const uint64_t cap = (uint64_t)n_cu; // one 1024-thread workgroup per CU (LDS-limited), persistent over tiles
		const unsigned g = (unsigned)(P.n_tiles < cap ? P.n_tiles : cap);
		hipError_t e;
		if (inverse) {
			e = func_lds_limit(reinterpret_cast<const void *>(&k_ntt_tiled<T, TW, true>), (int)lds);
			if (e != hipSuccess) return e;
			hipLaunchKernelGGL((k_ntt_tiled<T, TW, true>), dim3(g), dim3(kTiledThreads), lds, s, (T *)data, d_mul8, d_s_evals, P);
		} else {
			e = func_lds_limit(reinterpret_cast<const void *>(&k_ntt_tiled<T, TW, false>), (int)lds);
			if (e != hipSuccess) return e;
			hipLaunchKernelGGL((k_ntt_tiled<T, TW, false>), dim3(g), dim3(kTiledThreads), lds, s, (T *)data, d_mul8, d_s_evals, P);
		}
		e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

template <typename T>
static hipError_t run_tiled_tw(hipStream_t s, int n_cu, bool inverse, void *data, uint32_t tw_level, const uint8_t *d_mul8,
                               const uint64_t *d_s_evals, uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z,
                               uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds)
{
	switch (tw_level) {
	case 3: return run_tiled<T, 3>(s, n_cu, inverse, data, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 4: return run_tiled<T, 4>(s, n_cu, inverse, data, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 5: return run_tiled<T, 5>(s, n_cu, inverse, data, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	default: return hipErrorInvalidValue;
	}
}

// returns hipErrorNotSupported when the shape / field combination is left to the per-layer kernels
hipError_t launch_ntt_tiled(hipStream_t s, int n_cu, bool inverse, void *data, uint32_t elem_level, uint32_t tw_level,
                            const uint8_t *d_mul8, const uint64_t *d_s_evals, uint32_t log_domain, uint32_t log_x, uint32_t log_y,
                            uint32_t log_z, uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds)
{
	if (tw_level > 5 || tw_level < 3 || tw_level > elem_level) return hipErrorNotSupported;
	if (log_x + log_y + log_z < 12) return hipErrorNotSupported; // tiny transforms: per-layer kernels
	switch (elem_level) {
	case 4: return tw_level <= 4 ? run_tiled_tw<uint16_t>(s, n_cu, inverse, data, tw_level, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds) : hipErrorNotSupported;
	case 5: return run_tiled_tw<uint32_t>(s, n_cu, inverse, data, tw_level, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 6: return run_tiled_tw<uint64_t>(s, n_cu, inverse, data, tw_level, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	case 7: return run_tiled_tw<uint4>(s, n_cu, inverse, data, tw_level, d_mul8, d_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds);
	default: return hipErrorNotSupported;
	}
}

} // namespace bn
