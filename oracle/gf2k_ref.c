/*
 * oracle/gf2k_ref.c -- TEST INFRASTRUCTURE ONLY (see gf2k_ref.h).
 *
 * Follows crates/field/src/arch/portable/pairwise_recursive_arithmetic.rs:12-81
 * (mul / square / mul_alpha / invert_or_zero of the tower recursion) and
 * crates/field/src/binary_field.rs:361-412 (multiplication by a subfield element).
 */
#include "gf2k_ref.h"

#include <string.h>

static inline uint64_t mask_bits(int nbits) { return nbits >= 64 ? ~0ull : ((1ull << nbits) - 1); }

/* pairwise_recursive_arithmetic.rs:54-60  mul_alpha: (a0, a1) -> (a1, a0 + a1*alpha_{k-1}) */
uint64_t ref_gf_mul_alpha(uint64_t a, int level)
{
	if (level == 0)
		return a & 1;
	int h = 1 << (level - 1);
	uint64_t m = mask_bits(h);
	uint64_t a0 = a & m, a1 = (a >> h) & m;
	uint64_t z1 = ref_gf_mul_alpha(a1, level - 1);
	return a1 | ((a0 ^ z1) << h);
}

/* pairwise_recursive_arithmetic.rs:18-28 */
uint64_t ref_gf_mul_slow(uint64_t a, uint64_t b, int level)
{
	if (level == 0)
		return a & b & 1;
	int h = 1 << (level - 1);
	uint64_t m = mask_bits(h);
	uint64_t a0 = a & m, a1 = (a >> h) & m;
	uint64_t b0 = b & m, b1 = (b >> h) & m;
	uint64_t z0 = ref_gf_mul_slow(a0, b0, level - 1);
	uint64_t z2 = ref_gf_mul_slow(a1, b1, level - 1);
	uint64_t z0z2 = z0 ^ z2;
	uint64_t z1 = ref_gf_mul_slow(a0 ^ a1, b0 ^ b1, level - 1) ^ z0z2;
	uint64_t z2a = ref_gf_mul_alpha(z2, level - 1);
	return z0z2 | ((z1 ^ z2a) << h);
}

/* 64 KiB product table for T_3 (8-bit), built from the recursion above on first use.
 * Purely an evaluation speed-up of the same function; tests check table == recursion. */
static uint8_t g_mul8[256][256];
static int g_mul8_ready;

static void build_mul8(void)
{
	for (int a = 0; a < 256; a++)
		for (int b = a; b < 256; b++) {
			uint8_t p = (uint8_t)ref_gf_mul_slow((uint64_t)a, (uint64_t)b, 3);
			g_mul8[a][b] = p;
			g_mul8[b][a] = p;
		}
	g_mul8_ready = 1;
}

uint64_t ref_gf_mul(uint64_t a, uint64_t b, int level)
{
	if (level < 3)
		return ref_gf_mul_slow(a, b, level);
	if (level == 3) {
		if (!g_mul8_ready)
			build_mul8();
		return g_mul8[a & 0xff][b & 0xff];
	}
	int h = 1 << (level - 1);
	uint64_t m = mask_bits(h);
	uint64_t a0 = a & m, a1 = (a >> h) & m;
	uint64_t b0 = b & m, b1 = (b >> h) & m;
	uint64_t z0 = ref_gf_mul(a0, b0, level - 1);
	uint64_t z2 = ref_gf_mul(a1, b1, level - 1);
	uint64_t z0z2 = z0 ^ z2;
	uint64_t z1 = ref_gf_mul(a0 ^ a1, b0 ^ b1, level - 1) ^ z0z2;
	uint64_t z2a = ref_gf_mul_alpha(z2, level - 1);
	return z0z2 | ((z1 ^ z2a) << h);
}

/* pairwise_recursive_arithmetic.rs:38-44 */
uint64_t ref_gf_square(uint64_t a, int level)
{
	if (level == 0)
		return a & 1;
	int h = 1 << (level - 1);
	uint64_t m = mask_bits(h);
	uint64_t a0 = a & m, a1 = (a >> h) & m;
	uint64_t z0 = ref_gf_square(a0, level - 1);
	uint64_t z2 = ref_gf_square(a1, level - 1);
	uint64_t z2a = ref_gf_mul_alpha(z2, level - 1);
	return (z0 ^ z2) | (z2a << h);
}

/* pairwise_recursive_arithmetic.rs:64-80 */
uint64_t ref_gf_invert(uint64_t a, int level)
{
	if (level == 0)
		return a & 1;
	int h = 1 << (level - 1);
	uint64_t m = mask_bits(h);
	uint64_t a0 = a & m, a1 = (a >> h) & m;
	uint64_t a0z1 = a0 ^ ref_gf_mul_alpha(a1, level - 1);
	uint64_t delta = ref_gf_mul(a0, a0z1, level - 1) ^ ref_gf_square(a1, level - 1);
	uint64_t delta_inv = ref_gf_invert(delta, level - 1);
	uint64_t inv0 = ref_gf_mul(delta_inv, a0z1, level - 1);
	uint64_t inv1 = ref_gf_mul(delta_inv, a1, level - 1);
	return inv0 | (inv1 << h);
}

/* ---- level 7: BinaryField128b = (lo: T_6, hi: T_6) ---- */

ref_b128 ref_b128_mul(ref_b128 a, ref_b128 b)
{
	uint64_t z0 = ref_gf_mul(a.lo, b.lo, 6);
	uint64_t z2 = ref_gf_mul(a.hi, b.hi, 6);
	uint64_t z0z2 = z0 ^ z2;
	uint64_t z1 = ref_gf_mul(a.lo ^ a.hi, b.lo ^ b.hi, 6) ^ z0z2;
	uint64_t z2a = ref_gf_mul_alpha(z2, 6);
	ref_b128 r = {z0z2, z1 ^ z2a};
	return r;
}

ref_b128 ref_b128_square(ref_b128 a)
{
	uint64_t z0 = ref_gf_square(a.lo, 6);
	uint64_t z2 = ref_gf_square(a.hi, 6);
	ref_b128 r = {z0 ^ z2, ref_gf_mul_alpha(z2, 6)};
	return r;
}

ref_b128 ref_b128_mul_alpha(ref_b128 a)
{
	ref_b128 r = {a.hi, a.lo ^ ref_gf_mul_alpha(a.hi, 6)};
	return r;
}

ref_b128 ref_b128_invert(ref_b128 a)
{
	uint64_t a0z1 = a.lo ^ ref_gf_mul_alpha(a.hi, 6);
	uint64_t delta = ref_gf_mul(a.lo, a0z1, 6) ^ ref_gf_square(a.hi, 6);
	uint64_t delta_inv = ref_gf_invert(delta, 6);
	ref_b128 r = {ref_gf_mul(delta_inv, a0z1, 6), ref_gf_mul(delta_inv, a.hi, 6)};
	return r;
}

ref_b128 ref_b128_pow(ref_b128 a, uint64_t e)
{
	ref_b128 r = ref_b128_one();
	for (int i = 63; i >= 0; i--) {
		r = ref_b128_square(r);
		if ((e >> i) & 1)
			r = ref_b128_mul(r, a);
	}
	return r;
}

/* binary_field.rs:361-412: (a, b) = self.into(); (a*rhs, b*rhs) recursively, i.e. the subfield
 * scalar multiplies every 2^iota-bit limb; iota == 0 is a bit mask. */
ref_b128 ref_b128_mul_subfield(ref_b128 a, ref_b128 s, int iota)
{
	if (iota >= 7)
		return ref_b128_mul(a, s);
	if (iota == 0) {
		uint64_t m = 0 - (s.lo & 1);
		ref_b128 r = {a.lo & m, a.hi & m};
		return r;
	}
	int w = 1 << iota;
	uint64_t m = mask_bits(w);
	uint64_t sv = s.lo & m;
	ref_b128 r = {0, 0};
	for (int sh = 0; sh < 64; sh += w) {
		r.lo |= ref_gf_mul((a.lo >> sh) & m, sv, iota) << sh;
		r.hi |= ref_gf_mul((a.hi >> sh) & m, sv, iota) << sh;
	}
	return r;
}

void ref_b128_mul_p(const ref_b128 *a, const ref_b128 *b, ref_b128 *out) { *out = ref_b128_mul(*a, *b); }
void ref_b128_square_p(const ref_b128 *a, ref_b128 *out) { *out = ref_b128_square(*a); }
void ref_b128_invert_p(const ref_b128 *a, ref_b128 *out) { *out = ref_b128_invert(*a); }
void ref_b128_mul_subfield_p(const ref_b128 *a, const ref_b128 *s, int iota, ref_b128 *out)
{
	*out = ref_b128_mul_subfield(*a, *s, iota);
}
void ref_b128_mul_vec(const ref_b128 *a, const ref_b128 *b, ref_b128 *out, size_t n)
{
	for (size_t i = 0; i < n; i++)
		out[i] = ref_b128_mul(a[i], b[i]);
}

void ref_splitmix_fill(uint64_t seed, uint64_t *out, size_t n_words)
{
	uint64_t x = seed;
	for (size_t i = 0; i < n_words; i++) {
		x += 0x9E3779B97F4A7C15ull;
		uint64_t z = x;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		out[i] = z ^ (z >> 31);
	}
}
