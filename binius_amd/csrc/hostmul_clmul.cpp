// binius_amd/csrc/hostmul_clmul.cpp -- GF(2^128) tower product for HOST scalars through PCLMULQDQ.
//
// The tower basis has no carry-less-multiply structure, but the field is the field: pick an element beta of the tower whose
// powers 1, beta, .., beta^127 are a basis, and Phi : tower coordinates -> coordinates in that power basis is an F2-linear
// isomorphism onto GF(2)[x] / m(x), m = the minimal polynomial of beta.  a * b = Phi^-1( Phi(a) (x) Phi(b) mod m ):
//   * Phi, Phi^-1: 16 byte-indexed tables of 128-bit entries each (2 x 64 KiB), built at start-up from the powers of beta
//     (tower products by hostmul.hpp's table Karatsuba) and the inverse of their 128 x 128 bit matrix;
//   * the product: four PCLMULQDQ; the reduction modulo the (dense) m by Barrett with mu = floor(x^256 / m) -- exact in
//     GF(2)[x] --: q = H + floor(H mu0 / x^128), r = L + low128(q m0): eight more PCLMULQDQ.
// ~50 ns against ~200 ns for the 81 table look-ups of the Karatsuba form; a small sumcheck round costs its caller and the
// backend about a dozen of these.  Nothing is assumed about beta or m beyond what is checked here: the inverse matrix must
// exist, and the finished multiplier is compared with the table product on fixed and pseudo-random operands before it is
// used; on a host without PCLMULQDQ (or if any check fails) hostmul.hpp keeps the table form.
//
// The same state serves the HOST TAIL of a sumcheck (abi_kernels.cpp): the last rounds, on arrays of a few hundred elements,
// run on the host in the power basis (the device hands the arrays over already mapped through Phi: hostpoly_phi_nibble_table)
// -- sums of products accumulate the unreduced 256-bit carry-less products and are reduced once (hostpoly_round_sums).
//
// Host-only translation unit.  Built without any -m flag: only the functions that use the intrinsics carry the target
// attribute, and they are entered only after the cpuid check; on a host that is not x86-64 the unit compiles to stubs that
// report "not available" (hostmul.hpp then keeps the table form and the host tail stays off).
#include <stdint.h>

#include <cstdlib>
#include <cstring>

#include "hostmul.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#define BN_CLMUL_FN __attribute__((target("pclmul,sse4.1")))

namespace bn {

namespace {

struct u128 {
	uint64_t lo, hi;
};
inline u128 x128(u128 a, u128 b) { return u128{a.lo ^ b.lo, a.hi ^ b.hi}; }

struct clmul_state {
	bool ok = false;
	u128 fwd[16][256]; // Phi of (byte << 8k)
	u128 inv[16][256]; // Phi^-1 of (byte << 8k)
	u128 m0{0, 0};     // m(x) = x^128 + m0(x)
	u128 mu0{0, 0};    // floor(x^256 / m(x)) = x^128 + mu0(x)
};

inline u128 apply(const u128 (*tab)[256], u128 v)
{
	u128 r{0, 0};
	for (int k = 0; k < 8; k++) {
		r = x128(r, tab[k][(v.lo >> (8 * k)) & 0xFF]);
		r = x128(r, tab[8 + k][(v.hi >> (8 * k)) & 0xFF]);
	}
	return r;
}

// 128 x 128 -> 256 bit carry-less product
BN_CLMUL_FN inline void clmul256(u128 a, u128 b, u128 &lo, u128 &hi)
{
	const __m128i va = _mm_set_epi64x((long long)a.hi, (long long)a.lo), vb = _mm_set_epi64x((long long)b.hi, (long long)b.lo);
	const __m128i p00 = _mm_clmulepi64_si128(va, vb, 0x00), p11 = _mm_clmulepi64_si128(va, vb, 0x11);
	const __m128i mid = _mm_xor_si128(_mm_clmulepi64_si128(va, vb, 0x10), _mm_clmulepi64_si128(va, vb, 0x01));
	const __m128i l = _mm_xor_si128(p00, _mm_slli_si128(mid, 8)), h = _mm_xor_si128(p11, _mm_srli_si128(mid, 8));
	lo = u128{(uint64_t)_mm_cvtsi128_si64(l), (uint64_t)_mm_extract_epi64(l, 1)};
	hi = u128{(uint64_t)_mm_cvtsi128_si64(h), (uint64_t)_mm_extract_epi64(h, 1)};
}

// (L + x^128 H) mod m, Barrett: q = H + floor(H mu0 / x^128), r = L + low128(q m0)
BN_CLMUL_FN inline u128 reduce256(const clmul_state &st, u128 L, u128 H)
{
	u128 t_lo, t_hi;
	clmul256(H, st.mu0, t_lo, t_hi);
	const u128 q = x128(H, t_hi); // floor(H mu / x^128)
	clmul256(q, st.m0, t_lo, t_hi);
	return x128(L, t_lo);
}

BN_CLMUL_FN inline u128 mul_poly(const clmul_state &st, u128 a, u128 b)
{
	u128 L, H;
	clmul256(a, b, L, H);
	return reduce256(st, L, H);
}

// bit i of the 128-bit row vector
inline bool bit(const u128 &v, int i) { return ((i < 64 ? v.lo >> i : v.hi >> (i - 64)) & 1) != 0; }
inline void flip(u128 &v, int i)
{
	if (i < 64)
		v.lo ^= 1ull << i;
	else
		v.hi ^= 1ull << (i - 64);
}

BN_CLMUL_FN bool build_checked(clmul_state &st);
bool build(clmul_state &st)
{
	if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
	return build_checked(st);
}
BN_CLMUL_FN bool build_checked(clmul_state &st)
{
	for (uint64_t attempt = 1; attempt <= 8; attempt++) {
		// candidate generator: fixed odd constants, no structure needed -- almost every element has a minimal polynomial of degree 128
		const f128 beta{0x9E3779B97F4A7C15ull * attempt ^ 0x2545F4914F6CDD1Dull, 0xD1B54A32D192ED03ull * attempt ^ 0x8CB92BA72F3D8DD7ull};
		f128 pw[129];
		pw[0] = f128_one();
		for (int i = 1; i <= 128; i++) pw[i] = mul_host_table(pw[i - 1], beta);
		// invert M (column i = tower coordinates of beta^i): row-reduce [M^T | I] -- row i starts as (pw[i] | e_i); after the
		// elimination row j reads (e_j | c_j) with sum_i c_j[i] beta^i = 2^j, i.e. c_j = Phi(2^j)
		u128 left[128], right[128];
		for (int i = 0; i < 128; i++) {
			left[i] = u128{pw[i].lo, pw[i].hi};
			right[i] = u128{0, 0};
			flip(right[i], i);
		}
		bool singular = false;
		for (int col = 0; col < 128 && !singular; col++) {
			int piv = -1;
			for (int r = col; r < 128; r++)
				if (bit(left[r], col)) {
					piv = r;
					break;
				}
			if (piv < 0) {
				singular = true;
				break;
			}
			if (piv != col) {
				const u128 a = left[piv], b = right[piv];
				left[piv] = left[col];
				right[piv] = right[col];
				left[col] = a;
				right[col] = b;
			}
			for (int r = 0; r < 128; r++)
				if (r != col && bit(left[r], col)) {
					left[r] = x128(left[r], left[col]);
					right[r] = x128(right[r], right[col]);
				}
		}
		if (singular) continue;
		// byte tables
		for (int k = 0; k < 16; k++)
			for (int b = 0; b < 256; b++) {
				u128 f{0, 0}, g{0, 0};
				for (int t = 0; t < 8; t++)
					if ((b >> t) & 1) {
						f = x128(f, right[8 * k + t]);                              // Phi(2^(8k+t))
						g = x128(g, u128{pw[8 * k + t].lo, pw[8 * k + t].hi});      // Phi^-1(x^(8k+t)) = beta^(8k+t)
					}
				st.fwd[k][b] = f;
				st.inv[k][b] = g;
			}
		st.m0 = apply(st.fwd, u128{pw[128].lo, pw[128].hi}); // x^128 = m0(x) mod m
		// mu = floor(x^256 / m): long division, one quotient bit per step; rem = current remainder (degree < 128) of x^k
		{
			u128 mu{0, 0}, rem = st.m0; // x^128 mod m = m0, quotient bit 128 (the leading one) is implicit
			for (int k = 127; k >= 0; k--) { // rem * x: if its x^128 coefficient is set, the quotient has bit k and rem ^= m
				const bool top = (rem.hi >> 63) & 1;
				rem = u128{rem.lo << 1, (rem.hi << 1) | (rem.lo >> 63)};
				if (top) {
					rem = x128(rem, st.m0);
					flip(mu, k);
				}
			}
			st.mu0 = mu;
		}
		// the finished multiplier against the table product
		bool good = true;
		uint64_t s = 0x0123456789ABCDEFull;
		auto next = [&s] {
			s += 0x9E3779B97F4A7C15ull;
			uint64_t z = s;
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			return z ^ (z >> 31);
		};
		for (int t = 0; t < 256 && good; t++) {
			f128 a{next(), next()}, b{next(), next()};
			if (t == 0) a = f128{0, 0};
			if (t == 1) a = f128_one();
			if (t == 2) a = b = f128{~0ull, ~0ull};
			if (t == 3) a = f128{0, 1ull << 63}, b = f128{0, 1ull << 63};
			if (t >= 4 && t < 4 + 128) a = f128{t - 4 < 64 ? 1ull << (t - 4) : 0, t - 4 >= 64 ? 1ull << (t - 68) : 0};
			const u128 p = apply(st.inv, mul_poly(st, apply(st.fwd, u128{a.lo, a.hi}), apply(st.fwd, u128{b.lo, b.hi})));
			const f128 want = mul_host_table(a, b);
			good = p.lo == want.lo && p.hi == want.hi;
		}
		if (good) return true;
	}
	return false;
}

const clmul_state &state()
{
	static const clmul_state *st = [] {
		clmul_state *s = new clmul_state();
		s->ok = build(*s);
		return s;
	}();
	return *st;
}

} // namespace

bool hostmul_clmul_available() { return state().ok; }

namespace {
BN_CLMUL_FN f128 mul_host_clmul_impl(const clmul_state &st, f128 a, f128 b)
{
	const u128 p = apply(st.inv, mul_poly(st, apply(st.fwd, u128{a.lo, a.hi}), apply(st.fwd, u128{b.lo, b.hi})));
	return f128{p.lo, p.hi};
}
BN_CLMUL_FN void fold_impl(const clmul_state &st, hp128 *x, size_t half, hp128 z)
{
	const u128 zz{z.lo, z.hi};
	for (size_t i = 0; i < half; i++) {
		const u128 d = mul_poly(st, zz, u128{x[i].lo ^ x[i + half].lo, x[i].hi ^ x[i + half].hi});
		x[i].lo ^= d.lo;
		x[i].hi ^= d.hi;
	}
}
BN_CLMUL_FN void round_sums_impl(const clmul_state &st, const hp128 *a, const hp128 *b, size_t half, hp128 *y1, hp128 *yinf)
{
	// reduction modulo m is linear: the 256-bit carry-less products are XORed unreduced and reduced once per sum
	u128 L1{0, 0}, H1{0, 0}, Li{0, 0}, Hi{0, 0};
	for (size_t i = 0; i < half; i++) {
		u128 l, h;
		const u128 ah{a[half + i].lo, a[half + i].hi}, bh{b[half + i].lo, b[half + i].hi};
		clmul256(ah, bh, l, h);
		L1 = x128(L1, l);
		H1 = x128(H1, h);
		clmul256(u128{a[i].lo ^ ah.lo, a[i].hi ^ ah.hi}, u128{b[i].lo ^ bh.lo, b[i].hi ^ bh.hi}, l, h);
		Li = x128(Li, l);
		Hi = x128(Hi, h);
	}
	const u128 r1 = reduce256(st, L1, H1), ri = reduce256(st, Li, Hi);
	*y1 = hp128{r1.lo, r1.hi};
	*yinf = hp128{ri.lo, ri.hi};
}
// ---- the same two loops four elements at a time on VPCLMULQDQ (AVX-512): per 128-bit lane exactly the arithmetic above
#define BN_VCLMUL_FN __attribute__((target("avx512f,avx512bw,vpclmulqdq,pclmul,sse4.1")))
BN_VCLMUL_FN inline void vclmul256(__m512i a, __m512i b, __m512i &lo, __m512i &hi)
{
	const __m512i p00 = _mm512_clmulepi64_epi128(a, b, 0x00), p11 = _mm512_clmulepi64_epi128(a, b, 0x11);
	const __m512i mid = _mm512_xor_si512(_mm512_clmulepi64_epi128(a, b, 0x10), _mm512_clmulepi64_epi128(a, b, 0x01));
	lo = _mm512_xor_si512(p00, _mm512_bslli_epi128(mid, 8));
	hi = _mm512_xor_si512(p11, _mm512_bsrli_epi128(mid, 8));
}
BN_VCLMUL_FN inline __m512i vreduce256(__m512i L, __m512i H, __m512i mu0, __m512i m0)
{
	__m512i t_lo, t_hi;
	vclmul256(H, mu0, t_lo, t_hi);
	const __m512i q = _mm512_xor_si512(H, t_hi);
	vclmul256(q, m0, t_lo, t_hi);
	return _mm512_xor_si512(L, t_lo);
}
BN_VCLMUL_FN void fold_impl_v(const clmul_state &st, hp128 *x, size_t half, hp128 z)
{
	const __m512i zz = _mm512_broadcast_i32x4(_mm_set_epi64x((long long)z.hi, (long long)z.lo));
	const __m512i mu0 = _mm512_broadcast_i32x4(_mm_set_epi64x((long long)st.mu0.hi, (long long)st.mu0.lo));
	const __m512i m0 = _mm512_broadcast_i32x4(_mm_set_epi64x((long long)st.m0.hi, (long long)st.m0.lo));
	size_t i = 0;
	for (; i + 4 <= half; i += 4) {
		const __m512i a = _mm512_loadu_si512((const void *)(x + i)), b = _mm512_loadu_si512((const void *)(x + i + half));
		__m512i L, H;
		vclmul256(zz, _mm512_xor_si512(a, b), L, H);
		_mm512_storeu_si512((void *)(x + i), _mm512_xor_si512(a, vreduce256(L, H, mu0, m0)));
	}
	if (i < half) { // (half < 4, or a ragged end: the scalar form on what is left -- x[i + half] is addressed from the same base)
		const u128 zs{z.lo, z.hi};
		for (; i < half; i++) {
			const u128 d = mul_poly(st, zs, u128{x[i].lo ^ x[i + half].lo, x[i].hi ^ x[i + half].hi});
			x[i].lo ^= d.lo;
			x[i].hi ^= d.hi;
		}
	}
}
BN_VCLMUL_FN inline u128 fold4(__m512i v) // XOR of the four 128-bit lanes
{
	alignas(64) uint64_t w[8];
	_mm512_store_si512((void *)w, v);
	return u128{w[0] ^ w[2] ^ w[4] ^ w[6], w[1] ^ w[3] ^ w[5] ^ w[7]};
}
BN_VCLMUL_FN void round_sums_impl_v(const clmul_state &st, const hp128 *a, const hp128 *b, size_t half, hp128 *y1, hp128 *yinf)
{
	__m512i L1 = _mm512_setzero_si512(), H1 = L1, Li = L1, Hi = L1;
	size_t i = 0;
	for (; i + 4 <= half; i += 4) {
		const __m512i al = _mm512_loadu_si512((const void *)(a + i)), ah = _mm512_loadu_si512((const void *)(a + half + i));
		const __m512i bl = _mm512_loadu_si512((const void *)(b + i)), bh = _mm512_loadu_si512((const void *)(b + half + i));
		__m512i l, h;
		vclmul256(ah, bh, l, h);
		L1 = _mm512_xor_si512(L1, l);
		H1 = _mm512_xor_si512(H1, h);
		vclmul256(_mm512_xor_si512(al, ah), _mm512_xor_si512(bl, bh), l, h);
		Li = _mm512_xor_si512(Li, l);
		Hi = _mm512_xor_si512(Hi, h);
	}
	// the four lanes of every accumulator, then whatever is left of the range, then ONE reduction per sum
	u128 l1 = fold4(L1), h1 = fold4(H1), li = fold4(Li), hi = fold4(Hi);
	for (; i < half; i++) {
		u128 l, h;
		const u128 ahs{a[half + i].lo, a[half + i].hi}, bhs{b[half + i].lo, b[half + i].hi};
		clmul256(ahs, bhs, l, h);
		l1 = x128(l1, l);
		h1 = x128(h1, h);
		clmul256(u128{a[i].lo ^ ahs.lo, a[i].hi ^ ahs.hi}, u128{b[i].lo ^ bhs.lo, b[i].hi ^ bhs.hi}, l, h);
		li = x128(li, l);
		hi = x128(hi, h);
	}
	const u128 r1 = reduce256(st, l1, h1), ri = reduce256(st, li, hi);
	*y1 = hp128{r1.lo, r1.hi};
	*yinf = hp128{ri.lo, ri.hi};
}
bool have_vclmul()
{
	static const bool v = [] {
		const char *e = getenv("BN_HOSTMUL_VECTOR");
		if (e && e[0] == '0') return false;
		return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("vpclmulqdq");
	}();
	return v;
}

BN_CLMUL_FN hp128 mul_impl(const clmul_state &st, hp128 a, hp128 b)
{
	const u128 p = mul_poly(st, u128{a.lo, a.hi}, u128{b.lo, b.hi});
	return hp128{p.lo, p.hi};
}
} // namespace

f128 mul_host_clmul(f128 a, f128 b) { return mul_host_clmul_impl(state(), a, b); }

// ---- arithmetic in the power basis (the host tail of a sumcheck)
bool hostpoly_available() { return state().ok; }
hp128 hostpoly_from_tower(f128 v)
{
	const u128 p = apply(state().fwd, u128{v.lo, v.hi});
	return hp128{p.lo, p.hi};
}
f128 hostpoly_to_tower(hp128 v)
{
	const u128 p = apply(state().inv, u128{v.lo, v.hi});
	return f128{p.lo, p.hi};
}
hp128 hostpoly_mul(hp128 a, hp128 b) { return mul_impl(state(), a, b); }
bool hostpoly_vectorized() { return state().ok && have_vclmul(); }
void hostpoly_fold(hp128 *x, size_t half, hp128 z)
{
	if (half >= 4 && have_vclmul())
		fold_impl_v(state(), x, half, z);
	else
		fold_impl(state(), x, half, z);
}
void hostpoly_round_sums(const hp128 *a, const hp128 *b, size_t half, hp128 *y1, hp128 *yinf)
{
	if (half >= 4 && have_vclmul())
		round_sums_impl_v(state(), a, b, half, y1, yinf);
	else
		round_sums_impl(state(), a, b, half, y1, yinf);
}
void hostpoly_phi_nibble_table(uint64_t *out)
{
	// out[(16 p + e) * 2 ..] = Phi(e << 4 p): the layout of ctable.hpp's T (nibble position p, entry e)
	const clmul_state &st = state();
	for (int p = 0; p < 32; p++)
		for (int e = 0; e < 16; e++) {
			const u128 v = st.fwd[p >> 1][(p & 1) ? (e << 4) : e];
			out[2 * (16 * p + e)] = v.lo;
			out[2 * (16 * p + e) + 1] = v.hi;
		}
}

void hostpoly_phi_inv_nibble_table(uint64_t *out)
{
	const clmul_state &st = state();
	for (int p = 0; p < 32; p++)
		for (int e = 0; e < 16; e++) {
			const u128 v = st.inv[p >> 1][(p & 1) ? (e << 4) : e];
			out[2 * (16 * p + e)] = v.lo;
			out[2 * (16 * p + e) + 1] = v.hi;
		}
}

} // namespace bn

#else // not x86-64: no carry-less multiply route

namespace bn {
bool hostmul_clmul_available() { return false; }
f128 mul_host_clmul(f128 a, f128 b) { return mul_host_table(a, b); }
bool hostpoly_available() { return false; }
bool hostpoly_vectorized() { return false; }
hp128 hostpoly_from_tower(f128 v) { return hp128{v.lo, v.hi}; }
f128 hostpoly_to_tower(hp128 v) { return f128{v.lo, v.hi}; }
hp128 hostpoly_mul(hp128 a, hp128) { return a; }
void hostpoly_fold(hp128 *, size_t, hp128) {}
void hostpoly_round_sums(const hp128 *, const hp128 *, size_t, hp128 *, hp128 *) {}
void hostpoly_phi_nibble_table(uint64_t *) {}
void hostpoly_phi_inv_nibble_table(uint64_t *) {}
} // namespace bn

#endif
