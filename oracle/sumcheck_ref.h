/*
 * oracle/sumcheck_ref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's v3 bivariate-product sumcheck prover loop
 *   crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:133-254 (execute/fold/finish)
 *   crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-424 (calculate_round_evals,
 *   calculate_round_coeffs_from_evals)
 * with the Fiat-Shamir transcript replaced by a caller-supplied challenge list (the transcript
 * is host-side protocol bookkeeping outside the hot path).
 *
 * Also holds the multi-threaded CPU port timed by bench.py's cpu_baseline leg (kind "port"):
 * the same two ops (round-eval, fold) chunked over threads the way FastCpuLayer's
 * process_kernels_chunks does (crates/fast_compute/src/layer.rs:213-297).
 */
#ifndef BINIUS_ORACLE_SUMCHECK_REF_H
#define BINIUS_ORACLE_SUMCHECK_REF_H

#include "gf2k_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One round-eval: y_1 = sum_c alpha^c sum_i hi_a[i]*hi_b[i]; y_inf likewise on lo+hi.
 * multilins[j] has 2^n_vars elements; comps = n_comps pairs of multilinear indices
 * (IndexComposition<BivariateProduct,2>); out = {y_1, y_inf}.  threads <= 1: scalar. */
int ref_round_evals(const ref_b128 *const *multilins, size_t m, unsigned n_vars, const uint32_t *comps,
                    size_t n_comps, ref_b128 batch_coeff, ref_b128 out[2], int threads);

/* In-place fold of 2^n_vars evals into the first 2^(n_vars-1): x[i] += (x[i+half]-x[i])*z */
int ref_fold_high(ref_b128 *evals, unsigned n_vars, ref_b128 z, int threads);

/* Full prover loop. multilins are MODIFIED (folded in place).  sums[n_comps] are the claimed
 * sums.  challenges[n_vars]; batch_coeff is used in every round (as BatchSumcheckProver does
 * for a single prover).  round_coeffs_out[3*n_vars] = c0,c1,c2 per round; final_evals_out[m]. */
int ref_bivariate_sumcheck_prove(ref_b128 *const *multilins, size_t m, unsigned n_vars, const uint32_t *comps,
                                 size_t n_comps, const ref_b128 *sums, ref_b128 batch_coeff,
                                 const ref_b128 *challenges, ref_b128 *round_coeffs_out,
                                 ref_b128 *final_evals_out, int threads);

/* MLE-check (eq-indicator sumcheck) prover for bivariate products,
 * crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:
 *   calculate_round_evals with the eq indicator as last composition variable   :391-520
 *   calculate_round_coeffs_from_evals (y_0 from the sum, alpha)               :375-389
 *   execute: prime polynomial -> round polynomial, times eq_ind_prefix_eval   :273-318
 *   fold: prefix eval *= eq(alpha_r, challenge), multilinears folded, eq indicator halved by
 *   add_assign of its halves                                                   :120-123,145-254,320-346
 *   finish: final evaluations + eq_ind_prefix_eval                             :348-372
 * multilins (2^n_vars each) and eq_ind (2^(n_vars-1), the tensor expansion of
 * eq_ind_challenges[0..n_vars-1)) are MODIFIED.  round_coeffs_out[4*n_vars] (degree-3 round
 * polynomials c0..c3), final_evals_out[m+1] (last = eq_ind_prefix_eval). */
int ref_round_evals_eq(const ref_b128 *const *multilins, size_t m, unsigned n_vars, const ref_b128 *eq_ind,
                       const uint32_t *comps, size_t n_comps, ref_b128 batch_coeff, ref_b128 out[2]);
int ref_bivariate_mlecheck_prove(ref_b128 *const *multilins, size_t m, unsigned n_vars, ref_b128 *eq_ind,
                                 const ref_b128 *eq_ind_challenges, const uint32_t *comps, size_t n_comps,
                                 const ref_b128 *sums, ref_b128 batch_coeff, const ref_b128 *challenges,
                                 ref_b128 *round_coeffs_out, ref_b128 *final_evals_out);

/* Evaluate the multilinear extension of evals (2^n_vars) at point (low variable first) --
 * crates/math/src/multilinear_extension.rs:163 via tensor expansion + inner product. */
ref_b128 ref_mle_evaluate(const ref_b128 *evals, unsigned n_vars, const ref_b128 *point);

/* crates/math/src/univariate.rs:264-270 Horner */
ref_b128 ref_evaluate_univariate(const ref_b128 *coeffs, size_t n, ref_b128 x);

#ifdef __cplusplus
}
#endif
#endif
