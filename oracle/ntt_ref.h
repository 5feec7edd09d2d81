/*
 * oracle/ntt_ref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's additive NTT (LCH14/DP24):
 *   twiddles   crates/ntt/src/twiddle.rs:141-168, 244-313
 *   subspace   crates/math/src/binary_subspace.rs:33-38 (basis beta_i = 1 << i)
 *   transform  crates/ntt/src/tests/reference.rs:68-160 (the reference's own scalar NTT)
 *   FRI fold   crates/ntt/src/fri.rs:27-245 and crates/compute/src/cpu/layer.rs:304-391
 *
 * The reference holds no known-answer vectors for the NTT; it pins it by the scalar reference
 * (restated here), forward/inverse round trip and agreement between variants
 * (crates/ntt/src/tests/ntt_tests.rs:24-186).  tests/test_oracle_ntt.py additionally checks
 * the mathematical definition (novel-basis polynomial evaluation) on small domains.
 */
#ifndef BINIUS_ORACLE_NTT_REF_H
#define BINIUS_ORACLE_NTT_REF_H

#include "gf2k_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

#define REF_NTT_MAX_DIM 64

/* Twiddle basis of the canonical subspace span{1<<0..1<<(d-1)} of T_level (level <= 6):
 * s_evals[i*REF_NTT_MAX_DIM + b] = normalized W_i(beta_{i+1+b}), b < d-1-i.
 * (OnTheFlyTwiddleAccess::generate, twiddle.rs:107-124 + precompute_subspace_evals :244-306) */
int ref_ntt_s_evals(int level, int log_domain, uint64_t *s_evals);

/* layer-i twiddle for index j: subset sum (twiddle.rs:141-143,163-168) */
uint64_t ref_ntt_twiddle(const uint64_t *s_evals, int log_domain, int layer, uint64_t index);

/* AdditiveNTT::get_subspace_eval(i, j) = s_evals[log_domain - i].get(j)
 * (crates/ntt/src/single_threaded.rs:91-93) */
uint64_t ref_ntt_get_subspace_eval(const uint64_t *s_evals, int log_domain, int i, uint64_t j);

/* data: 2^(log_x+log_y+log_z) elements of T_elem_level stored contiguously, elem_level in 3..7
 * (1,2,4,8,16 bytes each); twiddles live in T_tw_level (tw_level <= min(elem_level,6)).
 * Element (x,y,z) sits at x | y<<log_x | z<<(log_x+log_y); transform runs along y
 * (crates/ntt/src/additive_ntt.rs:8-27, tests/reference.rs:170-204). */
int ref_ntt_forward(void *data, int elem_level, int tw_level, const uint64_t *s_evals, int log_domain,
                    int log_x, int log_y, int log_z, uint64_t coset, int coset_bits, int skip_rounds);
int ref_ntt_inverse(void *data, int elem_level, int tw_level, const uint64_t *s_evals, int log_domain,
                    int log_x, int log_y, int log_z, uint64_t coset, int coset_bits, int skip_rounds);

/* ComputeLayerExecutor::fri_fold restated from cpu/layer.rs:304-391 */
int ref_fri_fold(const uint64_t *s_evals, int tw_level, int log_domain, int log_len, int log_batch_size,
                 const ref_b128 *challenges, size_t n_challenges, const ref_b128 *data_in, size_t in_len,
                 ref_b128 *data_out, size_t out_len);

/* binius_ntt::fri::fold_interleaved restated from ntt/src/fri.rs:27-245 -- the independent
 * formula the reference's test_generic_fri_fold compares against
 * (crates/compute_test_utils/src/layer.rs:568) */
int ref_fold_interleaved(const uint64_t *s_evals, int tw_level, int log_domain, int log_len, int log_batch_size,
                         const ref_b128 *challenges, size_t n_challenges, const ref_b128 *codeword,
                         size_t in_len, ref_b128 *out, size_t out_len);

#ifdef __cplusplus
}
#endif
#endif
