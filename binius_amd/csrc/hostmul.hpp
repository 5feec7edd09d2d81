// binius_amd/csrc/hostmul.hpp -- GF(2^128) tower product for HOST scalars (protocol scalars behind bn_scalar_mul, the
// coefficient algebra of the old-HAL routing): the tower recursion of pairwise_recursive_arithmetic.rs:18-28 as
// Karatsuba down to GF(2^8), whose 256 x 256 products come from a table built once from the bilinear walk
// (gf128.hpp mul_walk<3>).  81 table look-ups + ~600 word operations instead of the walk's 127 mulx steps:
// ~0.2 us instead of ~0.9 us (mul_host_table); on hosts with PCLMULQDQ mul_host takes the route of hostmul_clmul.cpp instead.  A small round of the sumcheck costs the caller three of these (evaluate_univariate,
// powers of the batching coefficient), which is a tenth of the round once the launch is off the critical path (arm.hpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <memory>

#include "gf128.hpp"

namespace bn {

struct hostmul_table {
	uint8_t t[256][256];
	hostmul_table()
	{
		for (unsigned a = 0; a < 256; a++)
			for (unsigned b = a; b < 256; b++) {
				const uint8_t p = (uint8_t)mul_walk<3>(f128{a, 0}, b).lo;
				t[a][b] = p;
				t[b][a] = p;
			}
	}
};
inline const hostmul_table &hostmul_tab()
{
	static const std::unique_ptr<hostmul_table> tab(new hostmul_table());
	return *tab;
}

// product in T_K, operands and result in the low 2^K bits (K = 3 .. 6)
template <int K>
inline uint64_t hostmul_k(const hostmul_table &tb, uint64_t a, uint64_t b)
{
	if constexpr (K == 3) {
		return tb.t[a & 0xFF][b & 0xFF];
	} else {
		constexpr int H = 1 << (K - 1);
		constexpr uint64_t M = (1ull << H) - 1;
		const uint64_t a0 = a & M, a1 = (a >> H) & M, b0 = b & M, b1 = (b >> H) & M;
		const uint64_t z0 = hostmul_k<K - 1>(tb, a0, b0);
		const uint64_t z2 = hostmul_k<K - 1>(tb, a1, b1);
		const uint64_t z1 = hostmul_k<K - 1>(tb, a0 ^ a1, b0 ^ b1);
		const uint64_t lo = z0 ^ z2;
		const uint64_t hi = (z1 ^ lo ^ mulx64<K - 2>(z2)) & M; // X_{K-1}^2 = X_{K-1} X_{K-2} + 1
		return lo | (hi << H);
	}
}

inline f128 mul_host_table(f128 a, f128 b)
{
	const hostmul_table &tb = hostmul_tab();
	const uint64_t z0 = hostmul_k<6>(tb, a.lo, b.lo);
	const uint64_t z2 = hostmul_k<6>(tb, a.hi, b.hi);
	const uint64_t z1 = hostmul_k<6>(tb, a.lo ^ a.hi, b.lo ^ b.hi);
	const uint64_t lo = z0 ^ z2;
	return f128{lo, z1 ^ lo ^ mulx64<5>(z2)};
}

// The same product through PCLMULQDQ in an isomorphic power basis (hostmul_clmul.cpp: ~50 ns); available after its start-up
// self-check against mul_host_table on a host that has the instruction.  BN_HOSTMUL=table keeps the table form.
bool hostmul_clmul_available();
f128 mul_host_clmul(f128 a, f128 b);

// ---- the same field in the power basis of hostmul_clmul.cpp (coordinates Phi(v)): products are four PCLMULQDQ plus a
// Barrett reduction, sums of products reduce once.  Used by the host tail of a sumcheck (abi_kernels.cpp); only meaningful
// when hostpoly_available().
struct hp128 {
	uint64_t lo, hi;
};
bool hostpoly_available();
bool hostpoly_vectorized(); // the folds and sums below run four elements per instruction (AVX-512 VPCLMULQDQ); BN_HOSTMUL_VECTOR=0: never
hp128 hostpoly_from_tower(f128 v);
f128 hostpoly_to_tower(hp128 v);
hp128 hostpoly_mul(hp128 a, hp128 b);
void hostpoly_fold(hp128 *x, size_t half, hp128 z); // x[i] += z (x[i] + x[i + half]), i < half
// y1 = sum a[half + i] b[half + i], yinf = sum (a[i] + a[half + i]) (b[i] + b[half + i]), i < half
void hostpoly_round_sums(const hp128 *a, const hp128 *b, size_t half, hp128 *y1, hp128 *yinf);
void hostpoly_phi_nibble_table(uint64_t *out); // [1024]: Phi(e << 4 p) at entry 16 p + e (the layout of ctable.hpp's T)
void hostpoly_phi_inv_nibble_table(uint64_t *out); // the same for the inverse map (power basis -> tower basis)

inline f128 mul_host(f128 a, f128 b)
{
	static const bool fast = [] {
		const char *e = getenv("BN_HOSTMUL");
		return !(e && e[0] == 't') && hostmul_clmul_available();
	}();
	return fast ? mul_host_clmul(a, b) : mul_host_table(a, b);
}

} // namespace bn
