/* oracle/fastcpu_ref.c -- an OPTIMIZED CPU version of the bivariate-product sumcheck loop, used only as the
 * `cpu_baseline` leg of bench.py and checked against the scalar restatement (oracle/sumcheck_ref.c) by
 * tests/test_oracle_fastcpu.py.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.
 *
 * BASELINE.md section 2 asks for the reference's optimized CPU strategy (FastCpuLayer,
 * crates/fast_compute/src/layer.rs:213-297, 515-550: chunked over the threads, the best field arithmetic the
 * host offers) next to the scalar CpuLayer restatement.  The reference gets its fast GF(2^128) products on
 * x86 from packed GFNI arithmetic in the tower basis (crates/field/src/arch/x86_64/packed_128.rs:92-99) and
 * ALSO ships the isomorphic POLYVAL field with carry-less-multiply arithmetic
 * (crates/field/src/polyval.rs:262-330 Montgomery multiplication, :516-784 the two 128 x 128 basis-change
 * matrices).  This file takes the second route, which needs no packed-field machinery: convert the inputs
 * once with the reference's BINARY_TO_POLYVAL_TRANSFORMATION (a ring isomorphism onto the Montgomery form:
 * phi(a*b) = phi(a) (x) phi(b), the reference's test_to_from_tower_basis), run every fold and every
 * round evaluation with PCLMULQDQ, convert the two round sums back.  The transcript is bit-identical to the
 * scalar restatement's.  The two matrices are DATA handed in by the caller (tests/golden/field_kats.json). */
#include <immintrin.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gf2k_ref.h"
#include "sumcheck_ref.h"

typedef struct {
	__m128i t[16][256]; /* byte-sliced 128 x 128 GF(2) matrix: image of byte p with value b */
} bytemat;

static void bytemat_build(bytemat *m, const ref_b128 *rows /* image of bit i, i < 128 */)
{
	for (int p = 0; p < 16; p++)
		for (int b = 0; b < 256; b++) {
			uint64_t lo = 0, hi = 0;
			for (int k = 0; k < 8; k++)
				if ((b >> k) & 1) {
					lo ^= rows[8 * p + k].lo;
					hi ^= rows[8 * p + k].hi;
				}
			m->t[p][b] = _mm_set_epi64x((long long)hi, (long long)lo);
		}
}

static inline __m128i bytemat_apply(const bytemat *m, __m128i x)
{
	uint8_t b[16];
	_mm_storeu_si128((__m128i *)b, x);
	__m128i r = _mm_setzero_si128();
	for (int p = 0; p < 16; p++) r = _mm_xor_si128(r, m->t[p][b[p]]);
	return r;
}

/* a (x) b = a * b * x^-128 mod x^128 + x^127 + x^126 + x^121 + 1 (polyval.rs:262-330) */
static inline __m128i mont_mul(__m128i a, __m128i b)
{
	const __m128i g = _mm_set_epi64x(0, (long long)0xC200000000000000ull); /* x^57 + x^62 + x^63 */
	/* 128 x 128 -> 256 carry-less product, limbs c0..c3 */
	__m128i lo = _mm_clmulepi64_si128(a, b, 0x00), hi = _mm_clmulepi64_si128(a, b, 0x11);
	__m128i mid = _mm_xor_si128(_mm_clmulepi64_si128(a, b, 0x10), _mm_clmulepi64_si128(a, b, 0x01));
	lo = _mm_xor_si128(lo, _mm_slli_si128(mid, 8));
	hi = _mm_xor_si128(hi, _mm_srli_si128(mid, 8));
	/* Montgomery reduction, 64 bits at a time: with P = 1 + g x^64 + x^128, adding c0 * P clears limb 0 */
	__m128i t = _mm_clmulepi64_si128(lo, g, 0x00);       /* c0 * g */
	lo = _mm_xor_si128(lo, _mm_slli_si128(t, 8));         /* c1 ^= lo64(c0 g) */
	hi = _mm_xor_si128(hi, _mm_srli_si128(t, 8));         /* c2 ^= hi64(c0 g) */
	hi = _mm_xor_si128(hi, _mm_move_epi64(lo));           /* c2 ^= c0 */
	t = _mm_clmulepi64_si128(lo, g, 0x01);                /* c1' * g */
	hi = _mm_xor_si128(hi, t);                            /* c2 ^= lo64, c3 ^= hi64 */
	hi = _mm_xor_si128(hi, _mm_slli_si128(_mm_srli_si128(lo, 8), 8)); /* c3 ^= c1' */
	return hi;
}

static inline __m128i ld(const ref_b128 *p) { return _mm_loadu_si128((const __m128i *)p); }
static inline void st(ref_b128 *p, __m128i v) { _mm_storeu_si128((__m128i *)p, v); }
static inline ref_b128 to_ref(__m128i v)
{
	ref_b128 r;
	_mm_storeu_si128((__m128i *)&r, v);
	return r;
}

/* returns 0, 1 on bad arguments, 2 if the host has no PCLMULQDQ.  multilins are converted in place to the
 * POLYVAL representation and folded there (their final contents are not meaningful to the caller). */
int ref_fast_bivariate_sumcheck_prove(ref_b128 *const *multilins, size_t m, unsigned n_vars, const uint32_t *comps, size_t n_comps,
                                      const ref_b128 *sums, ref_b128 batch_coeff, const ref_b128 *challenges,
                                      const ref_b128 *binary_to_polyval /*[128]*/, const ref_b128 *polyval_to_binary /*[128]*/,
                                      ref_b128 *round_coeffs_out, ref_b128 *final_evals_out, int threads)
{
	if (!__builtin_cpu_supports("pclmul")) return 2;
	if (threads < 1) threads = 1;
	bytemat *fwd = malloc(sizeof(bytemat)), *inv = malloc(sizeof(bytemat)); /* (glibc: 16-byte aligned) */
	if (!fwd || !inv) return 1;
	bytemat_build(fwd, binary_to_polyval);
	bytemat_build(inv, polyval_to_binary);
	for (size_t c = 0; c < n_comps; c++)
		if (comps[2 * c] >= m || comps[2 * c + 1] >= m) return 1;
	const size_t n = (size_t)1 << n_vars;
	for (size_t j = 0; j < m; j++) {
		ref_b128 *x = multilins[j];
#pragma omp parallel for num_threads(threads) schedule(static)
		for (size_t i = 0; i < n; i++) st(&x[i], bytemat_apply(fwd, ld(&x[i])));
	}
	/* powers of the batching coefficient, in the POLYVAL representation */
	__m128i *coeff = malloc(sizeof(__m128i) * (n_comps ? n_comps : 1));
	{
		ref_b128 c = ref_b128_one();
		for (size_t k = 0; k < n_comps; k++) {
			coeff[k] = bytemat_apply(fwd, ld(&c));
			c = ref_b128_mul(c, batch_coeff);
		}
	}
	ref_b128 batched_sum = ref_evaluate_univariate(sums, n_comps, batch_coeff);
	for (unsigned round = 0; round < n_vars; round++) {
		const unsigned rem = n_vars - round;
		const size_t half = (size_t)1 << (rem - 1);
		/* round evaluation: y_1 = sum_c coeff_c sum_i hi_a hi_b ; y_inf with lo + hi (bivariate_product.rs:303-408) */
		__m128i y1 = _mm_setzero_si128(), yinf = _mm_setzero_si128();
		for (size_t c = 0; c < n_comps; c++) {
			const ref_b128 *a = multilins[comps[2 * c]], *b = multilins[comps[2 * c + 1]];
			uint64_t s1lo = 0, s1hi = 0, silo = 0, sihi = 0;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(^ : s1lo, s1hi, silo, sihi)
			for (size_t i = 0; i < half; i++) {
				const __m128i al = ld(&a[i]), ah = ld(&a[half + i]), bl = ld(&b[i]), bh = ld(&b[half + i]);
				const __m128i p1 = mont_mul(ah, bh), pi = mont_mul(_mm_xor_si128(al, ah), _mm_xor_si128(bl, bh));
				s1lo ^= (uint64_t)_mm_cvtsi128_si64(p1);
				s1hi ^= (uint64_t)_mm_extract_epi64(p1, 1);
				silo ^= (uint64_t)_mm_cvtsi128_si64(pi);
				sihi ^= (uint64_t)_mm_extract_epi64(pi, 1);
			}
			y1 = _mm_xor_si128(y1, mont_mul(_mm_set_epi64x((long long)s1hi, (long long)s1lo), coeff[c]));
			yinf = _mm_xor_si128(yinf, mont_mul(_mm_set_epi64x((long long)sihi, (long long)silo), coeff[c]));
		}
		const ref_b128 ev1 = to_ref(bytemat_apply(inv, y1)), evinf = to_ref(bytemat_apply(inv, yinf));
		/* calculate_round_coeffs_from_evals (:410-424), in the tower basis like the scalar restatement */
		const ref_b128 c0 = ref_b128_add(batched_sum, ev1), c2 = evinf, c1 = ref_b128_add(ref_b128_add(ev1, c0), c2);
		ref_b128 *rc = &round_coeffs_out[3 * round];
		rc[0] = c0;
		rc[1] = c1;
		rc[2] = c2;
		batched_sum = ref_evaluate_univariate(rc, 3, challenges[round]);
		/* fold (:168-232): x0 += z (x1 - x0) */
		const __m128i z = bytemat_apply(fwd, ld(&challenges[round]));
		for (size_t j = 0; j < m; j++) {
			ref_b128 *x = multilins[j];
#pragma omp parallel for num_threads(threads) schedule(static)
			for (size_t i = 0; i < half; i++) {
				const __m128i x0 = ld(&x[i]), x1 = ld(&x[half + i]);
				st(&x[i], _mm_xor_si128(x0, mont_mul(_mm_xor_si128(x0, x1), z)));
			}
		}
	}
	for (size_t j = 0; j < m; j++) final_evals_out[j] = to_ref(bytemat_apply(inv, ld(&multilins[j][0])));
	free(coeff);
	free(fwd);
	free(inv);
	return 0;
}

/* sum_i a[i] * b[i] for two GF(2^128) vectors (the claimed sum of a bivariate-product sumcheck,
 * compute/src/cpu/layer.rs:226-246 for tower level 7), in the POLYVAL representation: phi is additive and
 * multiplicative, so phi^-1(XOR_i phi(a_i) (x) phi(b_i)) is the tower-basis sum.  The inputs are not modified.
 * Pinned to ref_inner_product by tests/test_oracle_fastcpu.py.  Returns 0, 1 on bad arguments, 2 without PCLMULQDQ. */
int ref_fast_inner_product(const ref_b128 *a, const ref_b128 *b, size_t n, const ref_b128 *binary_to_polyval /*[128]*/,
                           const ref_b128 *polyval_to_binary /*[128]*/, ref_b128 *out, int threads)
{
	if (!__builtin_cpu_supports("pclmul")) return 2;
	if (!a || !b || !out) return 1;
	if (threads < 1) threads = 1;
	bytemat *fwd = malloc(sizeof(bytemat)), *inv = malloc(sizeof(bytemat));
	if (!fwd || !inv) return 1;
	bytemat_build(fwd, binary_to_polyval);
	bytemat_build(inv, polyval_to_binary);
	uint64_t slo = 0, shi = 0;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(^ : slo, shi)
	for (size_t i = 0; i < n; i++) {
		const __m128i p = mont_mul(bytemat_apply(fwd, ld(&a[i])), bytemat_apply(fwd, ld(&b[i])));
		slo ^= (uint64_t)_mm_cvtsi128_si64(p);
		shi ^= (uint64_t)_mm_extract_epi64(p, 1);
	}
	*out = to_ref(bytemat_apply(inv, _mm_set_epi64x((long long)shi, (long long)slo)));
	free(fwd);
	free(inv);
	return 0;
}
