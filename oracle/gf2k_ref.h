/*
 * oracle/gf2k_ref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference's binary-tower field
 * arithmetic.  Nothing under binius_amd/ may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Parity pin: this restatement is checked (tests/test_oracle_field.py) against every
 * known-answer vector the reference holds for the path:
 *   - tower mul KATs  crates/field/src/binary_field.rs:931-1029
 *   - multiplicative generators  crates/field/src/binary_field.rs:740-747
 *   - the 128-entry BINARY_TO_POLYVAL_TRANSFORMATION table and the POLYVAL mul/square KAT
 *     crates/field/src/polyval.rs:516-646, 1112-1127  (forces the GF(2^128) product)
 *
 * Representation (crates/field/src/binary_field.rs:682-699, 740-764):
 *   level k field T_k has 2^k bits; an element of T_k is (lo, hi) with lo = low 2^(k-1) bits.
 *   T_k = T_{k-1}[X]/(X^2 + X*alpha_{k-1} + 1), alpha_0 = 1, alpha_{k-1} = X_{k-2}.
 *   BinaryField128b is one little-endian u128 == {lo64, hi64} in memory.
 */
#ifndef BINIUS_ORACLE_GF2K_REF_H
#define BINIUS_ORACLE_GF2K_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	uint64_t lo, hi;
} ref_b128;

/* levels 0..6: operands are the low 2^level bits of a uint64_t */
uint64_t ref_gf_mul(uint64_t a, uint64_t b, int level);
uint64_t ref_gf_mul_slow(uint64_t a, uint64_t b, int level); /* pure recursion, no tables */
uint64_t ref_gf_mul_alpha(uint64_t a, int level);
uint64_t ref_gf_square(uint64_t a, int level);
uint64_t ref_gf_invert(uint64_t a, int level); /* invert_or_zero */

ref_b128 ref_b128_mul(ref_b128 a, ref_b128 b);
ref_b128 ref_b128_square(ref_b128 a);
ref_b128 ref_b128_invert(ref_b128 a);
ref_b128 ref_b128_mul_alpha(ref_b128 a); /* multiply by X_6, the generator over T_6 */
ref_b128 ref_b128_pow(ref_b128 a, uint64_t e);
/* a * s where s is an element of T_iota (iota in 0..7) embedded in the low bits of `s` */
ref_b128 ref_b128_mul_subfield(ref_b128 a, ref_b128 s, int iota);

static inline ref_b128 ref_b128_add(ref_b128 a, ref_b128 b)
{
	ref_b128 r = {a.lo ^ b.lo, a.hi ^ b.hi};
	return r;
}
static inline ref_b128 ref_b128_zero(void)
{
	ref_b128 r = {0, 0};
	return r;
}
static inline ref_b128 ref_b128_one(void)
{
	ref_b128 r = {1, 0};
	return r;
}
static inline int ref_b128_eq(ref_b128 a, ref_b128 b) { return a.lo == b.lo && a.hi == b.hi; }

/* pointer-style wrappers for ctypes */
void ref_b128_mul_p(const ref_b128 *a, const ref_b128 *b, ref_b128 *out);
void ref_b128_square_p(const ref_b128 *a, ref_b128 *out);
void ref_b128_invert_p(const ref_b128 *a, ref_b128 *out);
void ref_b128_mul_subfield_p(const ref_b128 *a, const ref_b128 *s, int iota, ref_b128 *out);
void ref_b128_mul_vec(const ref_b128 *a, const ref_b128 *b, ref_b128 *out, size_t n);

/* SplitMix64 -- the documented PRNG for every synthetic input */
void ref_splitmix_fill(uint64_t seed, uint64_t *out, size_t n_words);

#ifdef __cplusplus
}
#endif
#endif
