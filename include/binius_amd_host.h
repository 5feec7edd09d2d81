/*
 * include/binius_amd_host.h -- C entry points of libbinius_amd_host.so, the COMPILED HOST MIRROR.
 *
 * This is NOT the drop-in boundary (that is include/binius_amd.h, which the reference's Rust host
 * binds directly).  The reference's callers of the HAL are Rust generics over ComputeLayer; with no
 * Rust toolchain in the build image they are mirrored in C++ (binius_amd/host/compute_layer.hpp,
 * sumcheck.hpp) and driven through these few calls, so that tests and bench.py run the prover loops
 * at compiled-host speed over the same C ABI:
 *
 *   bnh_bivariate_sumcheck_prove   BivariateSumcheckProver  execute/fold/finish loop
 *                                  crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:27-254
 *                                  (+ the multi-GPU variants of DESIGN.md section 6)
 *   bnh_bivariate_mlecheck_prove   BivariateMLEcheckProver   v3/bivariate_mlecheck.rs:27-372
 *   bnh_shm_*                      intra-node exchange of the per-round partials (host shared memory)
 *   bnh_rccl_*                     the same exchange through one ncclAllGather per round (librccl is
 *                                  bound at run time from the process's own copy)
 *
 * All functions return 0 on success or a BN_ERR_* code; bnh_last_error() describes the last failure
 * of the calling thread's library instance.
 */
#ifndef BINIUS_AMD_HOST_H
#define BINIUS_AMD_HOST_H

#include "binius_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *bnh_last_error(void);

/* combine callback of the generic sharded variant: XOR the rank partials left in d_partial (2
 * elements: y_1, y_inf) across ranks into evals[2]; 0 on success */
typedef int (*bnh_round_reduce_fn)(void *user, const void *d_partial, bn_f128 *evals);

/* One complete prove.  d_multilins[m]: 2^n_vars elements each, never modified; d_scratch: device
 * memory for the folded copies (m * 2^(n_vars-1) elements, + 64 with tail_rounds); comp_indices:
 * n_comps pairs; sums[n_comps]; challenges[n_vars (+ log2 world with tail_rounds)];
 * round_coeffs_out[3 * rounds]; final_evals_out[m].
 * Single GPU: reduce = NULL, rccl_comm = NULL, shm = NULL.
 * Sharded (rank holds the elements with index = rank mod world): exactly one of
 *   shm        handle from bnh_shm_open: partials combined in host shared memory; with tail_rounds != 0
 *              the residual log2(world) rounds run in the same call
 *   rccl_comm  communicator from bnh_rccl_init, d_partial (2 elements) and d_gathered (2 * world
 *              elements) device buffers: one ncclAllGather per round on the context's stream
 *   reduce     caller-supplied combine of d_partial */
int bnh_bivariate_sumcheck_prove(bn_ctx *ctx, uint32_t n_vars, uint32_t m, const void *const *d_multilins, void *d_scratch,
                                 uint64_t scratch_elems, uint32_t n_comps, const uint32_t *comp_indices, const bn_f128 *sums,
                                 const bn_f128 *batch_coeff, const bn_f128 *challenges, bn_f128 *round_coeffs_out,
                                 bn_f128 *final_evals_out, bnh_round_reduce_fn reduce, void *reduce_user, void *d_partial,
                                 void *rccl_comm, int world, void *d_gathered, void *shm, int tail_rounds);

/* d_eq_ind: 2^(n_vars-1) elements = tensor expansion of eq_ind_challenges[0 .. n_vars-1);
 * round_coeffs_out[4 * n_vars] (degree-3 round polynomials); final_evals_out[m + 1] (the last one is
 * eq_ind_prefix_eval); d_scratch: (m + 1) * 2^(n_vars-1) elements (the weighted prover needs 2^n_vars per weighted
 * multilinear and 2^(n_vars-1) per other one, and is not used when the scratch is smaller than that) */
int bnh_bivariate_mlecheck_prove(bn_ctx *ctx, uint32_t n_vars, uint32_t m, const void *const *d_multilins, const void *d_eq_ind,
                                 const bn_f128 *eq_ind_challenges, void *d_scratch, uint64_t scratch_elems, uint32_t n_comps,
                                 const uint32_t *comp_indices, const bn_f128 *sums, const bn_f128 *batch_coeff,
                                 const bn_f128 *challenges, bn_f128 *round_coeffs_out, bn_f128 *final_evals_out);

/* The same prover behind a handle, one call per SumcheckProver method (prove/batch_sumcheck.rs:38-70): for hosts whose
 * challenges come out of a transcript round by round -- what the Rust shim's Mi355xMLEcheckProver binds
 * (crates/binius_mi355x/src/mlecheck.rs).  execute: coeffs_out[4]; finish: final_evals_out[m + 1]. */
typedef struct bnh_mlecheck bnh_mlecheck;
int bnh_mlecheck_new(bn_ctx *ctx, uint32_t n_vars, uint32_t m, const void *const *d_multilins, const void *d_eq_ind,
                     const bn_f128 *eq_ind_challenges, void *d_scratch, uint64_t scratch_elems, uint32_t n_comps,
                     const uint32_t *comp_indices, const bn_f128 *sums, bnh_mlecheck **out);
int bnh_mlecheck_execute(bnh_mlecheck *prover, const bn_f128 *batch_coeff, bn_f128 *coeffs_out);
int bnh_mlecheck_fold(bnh_mlecheck *prover, const bn_f128 *challenge);
int bnh_mlecheck_finish(bnh_mlecheck *prover, bn_f128 *final_evals_out);
void bnh_mlecheck_free(bnh_mlecheck *prover);

/* which prover the calling thread's last bnh_mlecheck_new / bnh_bivariate_mlecheck_prove chose: 1 = WeightedMLEcheckProver (the indicator
 * carried inside one factor of every composition; plain bivariate rounds on the matrix cores), 0 = the literal mirror
 * BivariateMLEcheckProver (no proper 2-colouring of the compositions, an indicator coordinate equal to 0 or 1, too
 * little scratch, n_vars < 2, a table that is not the expansion of the coordinates, or BN_MLECHECK=eager), -1 = none yet */
int bnh_mlecheck_last_mode(void);

/* FRI commit phase (commit_interleaved, crates/core/src/protocols/fri/prove.rs:88-198), every fold round
 * (FRIFolder::execute_fold_round, :307-432) and finalize (:444-482) through the C++ mirror
 * binius_amd/host/fri.hpp, codewords and Merkle trees resident on the device.
 * d_message: 2^(log_dim + log_batch_size) elements; d_scratch: 2 * 2^(log_dim + log_batch_size + log_inv_rate) elements are always enough (codeword + folded codewords + trees)
 * ; challenges[log_dim + log_batch_size]; roots_out[(n_arities + 1) * 32]: the commitment, then one
 * root per committed oracle; terminate_out: 2^(log_inv_rate + n_final_challenges) elements or NULL;
 * phase_ms_out[2]: wall-clock ms of the commit and of the fold phase, or NULL. */
int bnh_fri_commit_fold(bn_ctx *ctx, uint32_t log_dim, uint32_t log_inv_rate, uint32_t log_batch_size, const uint32_t *fold_arities,
                        uint32_t n_arities, uint32_t n_test_queries, const void *d_message, void *d_scratch, uint64_t scratch_elems,
                        const bn_f128 *challenges, uint8_t *roots_out, bn_f128 *terminate_out, double *phase_ms_out);

/* The front-loaded batch prover (SumcheckBatchProver = protocols/sumcheck/prove/front_loaded.rs:33-203, BatchProver::run with the
 * transcript's samples handed in) over p BivariateSumcheckProvers on ONE layer, ascending by number of variables: per round
 * execute() on every live prover, one challenge, fold() on every live prover; a prover finishes in the round that equals its
 * number of variables.  prover_desc[3 i ..] = (n_vars, m, n_comps); d_multilins / comp_indices / sums: the provers' lists
 * concatenated; batch_coeffs[n_provers]; challenges[max n_vars]; d_scratch: sum over provers of m * 2^(n_vars - 1) elements.
 * round_proofs_out[2 * rounds]: the truncated round polynomials (RoundCoeffs::truncate, common.rs:101-105; missing coefficients
 * zero); final_evals_out[sum m]: the final evaluations in finishing order. */
int bnh_batch_sumcheck_prove(bn_ctx *ctx, uint32_t n_provers, const uint32_t *prover_desc, const void *const *d_multilins, const uint32_t *comp_indices,
                             const bn_f128 *sums, void *d_scratch, uint64_t scratch_elems, const bn_f128 *batch_coeffs, const bn_f128 *challenges,
                             bn_f128 *round_proofs_out, bn_f128 *final_evals_out);

/* piop::prove (crates/core/src/piop/prove.rs:148-395) through the C++ mirror binius_amd/host/piop.hpp: commit_interleaved of the
 * merged message (fri.hpp), one BivariateSumcheckProver per number of variables that has a committed multilinear, the
 * front-loaded batch prover interleaved with the FRI folder (prove_interleaved_fri_sumcheck, :306-395).
 *   committed_n_vars[n_committed] ascending, d_committed[i]: 2^n_vars elements (the packed committed multilinears, on the device)
 *   transparent_n_vars[n_transparent] ascending, d_transparent[i] likewise
 *   claims[3 i ..] = (n_vars, committed index, transparent index), claim_sums[i]          (PIOPSumcheckClaim, piop/verify.rs)
 *   d_message: 2^(log_dim + log_batch_size) elements = merge_multilins of the committed multilinears (piop/prove.rs:66-104);
 *              log_dim + log_batch_size must equal CommitMeta::total_vars
 *   d_scratch: codeword, folded codewords, Merkle trees and folded multilinears
 *   batch_coeffs: one per prover (sizes with a committed multilinear, ascending); challenges[total_vars]
 * The transcript comes back in writing order: items_out[2 i] = kind (0 round proof, 1 final evaluations of a finished prover,
 * 2 FRI round commitment, 3 FRI terminate codeword), items_out[2 i + 1] = its scalars (kinds 0, 1, 3: taken from scalars_out
 * in order) or 1 (kind 2: one 32-byte digest from digests_out).  phase_ms_out[2]: commit, prove (wall clock), or NULL. */
int bnh_piop_prove(bn_ctx *ctx, uint32_t n_committed, const uint32_t *committed_n_vars, const void *const *d_committed, uint32_t n_transparent,
                   const uint32_t *transparent_n_vars, const void *const *d_transparent, uint32_t n_claims, const uint32_t *claims, const bn_f128 *claim_sums,
                   uint32_t log_dim, uint32_t log_inv_rate, uint32_t log_batch_size, const uint32_t *fold_arities, uint32_t n_arities, uint32_t n_test_queries,
                   const void *d_message, void *d_scratch, uint64_t scratch_elems, const bn_f128 *batch_coeffs, uint32_t n_batch_coeffs,
                   const bn_f128 *challenges, uint32_t n_challenges, uint8_t *commitment_out, uint32_t *items_out, uint32_t max_items, uint32_t *n_items_out,
                   bn_f128 *scalars_out, uint64_t max_scalars, uint64_t *n_scalars_out, uint8_t *digests_out, uint32_t max_digests, uint32_t *n_digests_out,
                   double *phase_ms_out);

/* EqIndSumcheckProver (crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644) through the C++ mirror binius_amd/host/eq_ind.hpp,
 * over the old HAL (binius_hal::ComputationBackend: bn_hal_round_evals + the ComputeLayer's folds), evaluation order High-to-Low,
 * compositions of degree 1 .. 8 (degrees[c]; NULL: all 2; evaluation points 1 ..= degree: 1, infinity, the points 2, 3, ... of the default
 * interpolation domain, eq_ind.rs:664-668, math/src/univariate.rs:60-99): the zerocheck of a constraint set -- ONE composition per
 * constraint over ALL multilinears of the table (core/src/constraint_system/prove.rs:431-505).
 *   d_multilins[n_mls]: 2^n_vars elements each, FOLDED IN PLACE;  steps / steps_inf: the compositions and their leading forms
 *   (ArithCircuit::leading_term), concatenated, n_steps[c] / n_steps_inf[c] steps each;  sums[n_comps]: the claimed sums
 *   eq_ind_challenges[n_vars];  d_eq_ind: >= 2^(n_vars - 1) elements of scratch (the indicator's partial evaluations)
 *   round_coeffs_out[(D + 2) * n_vars], D = max(2, largest degree): the batched round polynomials (degree D + 1; the tables'
 *   constraints: D = 2, four coefficients per round);  final_evals_out[n_mls + 1]: the multilinears'
 *   evaluations at the challenges, then the indicator's prefix evaluation (eq_ind.rs:639-643) */
int bnh_eqind_sumcheck_prove(bn_ctx *ctx, uint32_t n_vars, uint32_t n_mls, void *const *d_multilins, uint32_t n_comps, const bn_step *steps,
                             const uint32_t *n_steps, const bn_step *steps_inf, const uint32_t *n_steps_inf, const uint32_t *degrees, const bn_f128 *sums,
                             const bn_f128 *eq_ind_challenges, void *d_eq_ind, uint64_t eq_ind_elems, const bn_f128 *batch_coeff, const bn_f128 *challenges,
                             bn_f128 *round_coeffs_out, bn_f128 *final_evals_out);

/* shared-memory exchange: rank 0 creates the segment `name` ("/..."), the others open it afterwards */
int bnh_shm_open(const char *name, int world, int rank, int create, void **handle_out);
int bnh_shm_close(void *handle);
/* every rank contributes n_words (<= 7) 64-bit words; out[world * n_words], rank-major */
int bnh_shm_allgather(void *handle, const uint64_t *in, uint32_t n_words, uint64_t *out);

/* RCCL, bound with dlopen from `librccl_path` (the librccl.so the process already uses) */
int bnh_rccl_open(const char *librccl_path);
int bnh_rccl_unique_id(void *out128);
int bnh_rccl_init(const void *id128, int world, int rank, void **comm_out);
int bnh_rccl_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif
