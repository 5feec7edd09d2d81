/*
 * oracle/hal_ref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's OLD hardware abstraction layer, `binius_hal::ComputationBackend`
 * (crates/hal/src/backend.rs:35-84) as implemented by `CpuBackend` (crates/hal/src/cpu.rs:20-90): the
 * sumcheck round calculation (crates/hal/src/sumcheck_round_calculation.rs:45-330, accesses :366-560) and
 * multilinear folding with small-field switchover (crates/hal/src/sumcheck_folding.rs:16-237), for scalar
 * (width-1) large-field vectors.  Used only by tests/ as the checker of bn_hal_* (include/binius_amd.h).
 */
#ifndef BINIUS_ORACLE_HAL_REF_H
#define BINIUS_ORACLE_HAL_REF_H

#include "layer_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { REF_ORDER_LOW_TO_HIGH = 0, REF_ORDER_HIGH_TO_LOW = 1 }; /* binius_math::EvaluationOrder */
enum { REF_HAL_ML_FOLDED = 0, REF_HAL_ML_TRANSPARENT = 1 };    /* hal/src/sumcheck_multilinear.rs:8-21 */

typedef struct {
	uint32_t kind;
	uint32_t tower_level;     /* TRANSPARENT: level of the packed subfield values */
	const ref_b128 *evals;    /* FOLDED: len large-field evaluations; TRANSPARENT: 2^n_vars_ml packed subfield values */
	uint64_t len;             /* FOLDED: stored evaluations, the rest of the cube equals suffix_eval; TRANSPARENT: ref_b128 words */
	ref_b128 suffix_eval;
	uint32_t n_vars_ml;       /* TRANSPARENT: variables of the multilinear (= n_vars + query variables) */
} ref_hal_multilinear;

typedef struct {
	const ref_step *composition;
	uint64_t n_steps;
	const ref_step *composition_at_infinity; /* ArithCircuit::leading_term, regular_sumcheck.rs:199-200 */
	uint64_t n_steps_inf;
	uint32_t eval_point_start, eval_point_end; /* SumcheckEvaluator::eval_point_indices */
	const ref_b128 *eq_ind;                    /* NULL (RegularSumcheckEvaluator) or 2^(n_vars-1) values (eq_ind.rs:676-704) */
} ref_hal_evaluator;

/* calculate_round_evals: out holds, evaluator by evaluator, one value per evaluation point index of its range
 * (0: X = 0, 1: X = 1, 2: X = infinity, 3 + k: nontrivial_points[k]).  tensor_query: the query EXPANSION
 * (2^query_vars values) used by TRANSPARENT multilinears; may be NULL when query_vars == 0. */
int ref_hal_round_evals(int order, uint32_t n_vars, const ref_b128 *tensor_query, uint32_t query_vars,
                        const ref_hal_multilinear *mls, uint32_t n_mls, const ref_hal_evaluator *evs, uint32_t n_evs,
                        const ref_b128 *nontrivial_points, uint32_t n_points, ref_b128 *out);

/* One multilinear of sumcheck_fold_multilinears.  FOLDED: single-variable lerp fold in the given order
 * (fold_right_lerp / fold_left_lerp_inplace, crates/math/src/fold.rs:528-576, 648-696).  TRANSPARENT (at its
 * switchover round): partial evaluation at the query (evaluate_partial_low / evaluate_partial_high,
 * crates/math/src/multilinear_extension.rs:253-341).  Writes *out_len evaluations to out (capacity out_cap). */
int ref_hal_fold_multilinear(int order, uint32_t n_vars, const ref_hal_multilinear *ml, ref_b128 challenge,
                             const ref_b128 *tensor_query, uint32_t query_vars, ref_b128 *out, uint64_t out_cap, uint64_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
