// binius_amd/csrc/gf128.hpp -- binary-tower field primitives shared by host and gfx950 device code.
//
// Arithmetic spec (what must be reproduced bit-for-bit):
//   tower  T_k = T_{k-1}[X_{k-1}] / (X_{k-1}^2 + X_{k-1}*alpha_{k-1} + 1), alpha_0 = 1,
//          alpha_{k-1} = X_{k-2}            crates/field/src/arch/portable/packed_arithmetic.rs:173-179
//   (lo, hi) split of an element             crates/field/src/binary_field.rs:682-699
//   mul / mul_alpha recursion                crates/field/src/arch/portable/pairwise_recursive_arithmetic.rs:12-62
//   BinaryField128b = one little-endian u128 crates/field/src/binary_field.rs:747
//
// Design note (not a translation of the reference's recursion): the F2-basis of the tower is the
// multilinear monomial basis in X_0..X_6 -- basis element 2^i equals prod_{k : bit k of i} X_k with
// X_k = 2^(2^k).  Multiplication by X_k acts independently on every 2^(k+1)-bit limb as
//     (l0, l1) -> (l1, l0 + l1 * X_{k-1})
// which is a handful of SWAR shifts/masks on a whole 128-bit word.  Everything here is built from
// that one primitive: z * 2^i is at most 7 applications of it, a constant multiplier becomes 32
// nibble tables (8 KiB of LDS, conflict-free by construction), and the generic product is a
// bilinear walk over those basis products.
#pragma once
#include <cstdlib>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>

#if defined(__HIPCC__)
#define BN_HD __host__ __device__
#else
#define BN_HD
#endif

namespace bn {
// Measurement knobs of rounds 1 - 4 whose A/B is settled (profiles/DEAD_ENDS.md): the shipped library runs their
// defaults and does not read them from the environment; a measurement build (tools/*.hip, -DBN_MEASUREMENT_KNOBS) still does.
// A shipped library that finds one of them SET says so once per knob on stderr -- an A/B script run against it would otherwise
// compare the default with itself and report "no difference" (ADVICE r5).
inline const char *settled_knob(const char *name)
{
#ifdef BN_MEASUREMENT_KNOBS
	return std::getenv(name);
#else
	if (std::getenv(name)) std::fprintf(stderr, "[binius_amd] %s is set but this build ignores the settled measurement knobs (build with `make BN_KNOBS=1`)\n", name);
	return nullptr;
#endif
}
} // namespace bn

namespace bn {

struct alignas(16) f128 {
	uint64_t lo, hi;
};

BN_HD inline f128 f128_zero() { return f128{0, 0}; }
BN_HD inline f128 f128_one() { return f128{1, 0}; }
BN_HD inline f128 operator^(f128 a, f128 b) { return f128{a.lo ^ b.lo, a.hi ^ b.hi}; }
BN_HD inline f128 &operator^=(f128 &a, f128 b)
{
	a.lo ^= b.lo;
	a.hi ^= b.hi;
	return a;
}
BN_HD inline bool operator==(f128 a, f128 b) { return a.lo == b.lo && a.hi == b.hi; }

// mask with the low half of every 2^(K+1)-bit limb set
template <int K>
BN_HD constexpr uint64_t lo_half_mask()
{
	uint64_t m = (K >= 6) ? ~0ull : ((1ull << (1u << (K < 6 ? K : 0))) - 1);
	for (int sh = 2 << K; sh < 64; sh <<= 1)
		m |= m << sh;
	return m;
}

// Multiply every 2^(K+1)-bit limb of a 64-bit word by X_K (K <= 5).
template <int K>
BN_HD inline uint64_t mulx64(uint64_t a)
{
	constexpr uint64_t M = lo_half_mask<K>();
	constexpr int H = 1 << K;
	uint64_t l0 = a & M;
	uint64_t l1 = (a >> H) & M;
	if constexpr (K == 0) {
		return l1 | ((l0 ^ l1) << 1); // alpha_0 = 1
	} else {
		return l1 | ((l0 ^ mulx64<K - 1>(l1)) << H);
	}
}

// Multiply a BinaryField128b element by X_K = 2^(2^K), K in 0..6.
template <int K>
BN_HD inline f128 mulx(f128 a)
{
	if constexpr (K == 6) {
		return f128{a.hi, a.lo ^ mulx64<5>(a.hi)};
	} else {
		return f128{mulx64<K>(a.lo), mulx64<K>(a.hi)};
	}
}

// z * 2^i  (i in 0..127): 2^i = prod_{k : bit k of i} X_k
BN_HD inline f128 mul_basis(f128 z, unsigned i)
{
	if (i & 1) z = mulx<0>(z);
	if (i & 2) z = mulx<1>(z);
	if (i & 4) z = mulx<2>(z);
	if (i & 8) z = mulx<3>(z);
	if (i & 16) z = mulx<4>(z);
	if (i & 32) z = mulx<5>(z);
	if (i & 64) z = mulx<6>(z);
	return z;
}

// Generic product as a bilinear walk: a*b with b in T_K (low 2^K bits of bw) =
//   P_{K-1}(a, b0) + P_{K-1}(a * X_{K-1}, b1).  Slow path only (tails, scalars, generic circuits);
// the hot kernels use nibble tables (constant multiplier) or the bit-sliced product.
template <int K>
BN_HD inline f128 mul_walk(f128 a, uint64_t bw)
{
	if constexpr (K == 0) {
		uint64_t m = 0 - (bw & 1);
		return f128{a.lo & m, a.hi & m};
	} else {
		constexpr int H = 1 << (K - 1);
		f128 r0 = mul_walk<K - 1>(a, bw);
		f128 r1 = mul_walk<K - 1>(mulx<K - 1>(a), bw >> H);
		return r0 ^ r1;
	}
}

BN_HD inline f128 mul_slow(f128 a, f128 b)
{
	f128 r0 = mul_walk<6>(a, b.lo);
	f128 r1 = mul_walk<6>(mulx<6>(a), b.hi);
	return r0 ^ r1;
}

// a * s, s an element of the subfield T_iota given in the low 2^iota bits of s
// (crates/field/src/binary_field.rs:361-412; the embedding T_iota -> T_7 is the identity on bits,
// so this equals the full product with the embedded element).
BN_HD inline f128 mul_subfield_slow(f128 a, uint64_t s, int iota)
{
	switch (iota) {
	case 0: return mul_walk<0>(a, s);
	case 1: return mul_walk<1>(a, s);
	case 2: return mul_walk<2>(a, s);
	case 3: return mul_walk<3>(a, s);
	case 4: return mul_walk<4>(a, s);
	case 5: return mul_walk<5>(a, s);
	default: return mul_walk<6>(a, s);
	}
}

BN_HD inline f128 square_slow(f128 a) { return mul_slow(a, a); }

BN_HD inline f128 pow_slow(f128 a, uint64_t e)
{
	f128 r = f128_one();
	for (int i = 63; i >= 0; i--) {
		r = mul_slow(r, r);
		if ((e >> i) & 1)
			r = mul_slow(r, a);
	}
	return r;
}

// Inverse in T_K (element in the low 2^K bits of x; 0 -> 0) by descending the tower: for
// a = a0 + a1 X over T_{K-1} with X^2 = X alpha + 1, the conjugate is (a0 + a1 alpha) + a1 X and the
// norm  a0 (a0 + a1 alpha) + a1^2  lies in T_{K-1}  (crates/field/src/arith_traits.rs InvertOrZero;
// the reference's portable tower inversion has the same shape).  ~600 walk steps for K = 7 against
// the 32k of a^(2^128 - 2) by square-and-multiply.
template <int K>
BN_HD inline uint64_t invert_tower64(uint64_t x)
{
	if constexpr (K == 0) {
		return x & 1;
	} else {
		constexpr int H = 1 << (K - 1);
		constexpr uint64_t M = (H >= 64) ? ~0ull : ((1ull << H) - 1);
		const uint64_t a0 = x & M, a1 = (x >> H) & M;
		uint64_t a1_alpha;
		if constexpr (K == 1)
			a1_alpha = a1;
		else
			a1_alpha = mulx64<K - 2>(a1) & M;
		const uint64_t t = a0 ^ a1_alpha;
		const uint64_t norm = mul_walk<K - 1>(f128{a0, 0}, t).lo ^ mul_walk<K - 1>(f128{a1, 0}, a1).lo;
		const uint64_t ninv = invert_tower64<K - 1>(norm & M);
		const uint64_t r0 = mul_walk<K - 1>(f128{t, 0}, ninv).lo & M;
		const uint64_t r1 = mul_walk<K - 1>(f128{a1, 0}, ninv).lo & M;
		return r0 | (r1 << H);
	}
}

BN_HD inline f128 invert_tower(f128 a)
{
	const uint64_t a0 = a.lo, a1 = a.hi;
	const uint64_t t = a0 ^ mulx64<5>(a1);
	const uint64_t norm = mul_walk<6>(f128{a0, 0}, t).lo ^ mul_walk<6>(f128{a1, 0}, a1).lo;
	const uint64_t ninv = invert_tower64<6>(norm);
	return f128{mul_walk<6>(f128{t, 0}, ninv).lo, mul_walk<6>(f128{a1, 0}, ninv).lo};
}

} // namespace bn
