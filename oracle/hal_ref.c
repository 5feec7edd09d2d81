/*
 * oracle/hal_ref.c -- TEST INFRASTRUCTURE ONLY (see hal_ref.h).
 */
#include "hal_ref.h"

#include <stdlib.h>
#include <string.h>

/* Value of a multilinear at index i of the CURRENT cube (2^n_vars points): Folded -> stored value or the constant
 * suffix (sumcheck_round_calculation.rs:421-441, 521-556); Transparent -> the partial evaluation at the tensor
 * query, materialised by the caller into `virt`. */
static ref_b128 ml_at(const ref_hal_multilinear *ml, const ref_b128 *virt, uint64_t i)
{
	if (ml->kind == REF_HAL_ML_FOLDED)
		return i < ml->len ? ml->evals[i] : ml->suffix_eval;
	return virt[i];
}

/* Transparent multilinear -> its 2^n_vars large-field values under the current query:
 * LowToHigh: evaluate_partial_low = fold_right; HighToLow: evaluate_partial_high = fold_left
 * (sumcheck_round_calculation.rs:404-418, 496-518; multilinear_extension.rs:253-341). */
static int materialise(int order, uint32_t n_vars, const ref_hal_multilinear *ml, const ref_b128 *tensor_query,
                       uint32_t query_vars, ref_b128 *out)
{
	if (ml->n_vars_ml != n_vars + query_vars) return 1;
	const ref_b128 one = {1, 0};
	const ref_b128 *q = query_vars ? tensor_query : &one;
	const size_t out_len = (size_t)1 << n_vars;
	if (order == REF_ORDER_LOW_TO_HIGH)
		return ref_fold_right(ml->evals, ml->len, (int)ml->tower_level, q, (size_t)1 << query_vars, out, out_len);
	return ref_fold_left(ml->evals, ml->len, (int)ml->tower_level, q, (size_t)1 << query_vars, out, out_len);
}

int ref_hal_round_evals(int order, uint32_t n_vars, const ref_b128 *tensor_query, uint32_t query_vars,
                        const ref_hal_multilinear *mls, uint32_t n_mls, const ref_hal_evaluator *evs, uint32_t n_evs,
                        const ref_b128 *nontrivial_points, uint32_t n_points, ref_b128 *out)
{
	if (n_vars == 0) return 1;
	/* union of the evaluation point ranges; the nontrivial points must cover indices 3.. (round_calculation.rs:113-125) */
	uint32_t pt_lo = 0, pt_hi = 0;
	for (uint32_t e = 0; e < n_evs; e++) {
		if (evs[e].eval_point_end < evs[e].eval_point_start) return 1;
		if (e == 0 || evs[e].eval_point_start < pt_lo) pt_lo = evs[e].eval_point_start;
		if (e == 0 || evs[e].eval_point_end > pt_hi) pt_hi = evs[e].eval_point_end;
	}
	if (n_points != (pt_hi > 3 ? pt_hi - 3 : 0)) return 2; /* Error::IncorrectNontrivialEvalPointsLength */

	const uint64_t half = (uint64_t)1 << (n_vars - 1);
	ref_b128 **virt = (ref_b128 **)calloc(n_mls ? n_mls : 1, sizeof(ref_b128 *));
	int rc = 0;
	for (uint32_t k = 0; k < n_mls && !rc; k++)
		if (mls[k].kind == REF_HAL_ML_TRANSPARENT) {
			virt[k] = (ref_b128 *)malloc(sizeof(ref_b128) << n_vars);
			rc = materialise(order, n_vars, &mls[k], tensor_query, query_vars, virt[k]);
		}
	size_t total = 0;
	for (uint32_t e = 0; e < n_evs; e++) total += evs[e].eval_point_end - evs[e].eval_point_start;
	memset(out, 0, total * sizeof(ref_b128));
	/* (one value per multilinear: a constraint set's zerocheck passes every column of its table, prove.rs:431-505) */
	ref_b128 *e0 = (ref_b128 *)calloc(3 * (size_t)(n_mls ? n_mls : 1), sizeof(ref_b128));
	ref_b128 *e1 = e0 + (n_mls ? n_mls : 1), *row = e1 + (n_mls ? n_mls : 1);
	for (uint64_t i = 0; i < half && !rc; i++) {
		for (uint32_t k = 0; k < n_mls; k++) {
			/* the substituted variable is the lowest one (LowToHigh: pairs 2i, 2i+1, round_calculation.rs:443-464)
			 * or the highest one (HighToLow: pairs i, i + 2^(n_vars-1), :521-556) */
			const uint64_t i0 = order == REF_ORDER_LOW_TO_HIGH ? 2 * i : i;
			const uint64_t i1 = order == REF_ORDER_LOW_TO_HIGH ? 2 * i + 1 : i + half;
			e0[k] = ml_at(&mls[k], virt[k], i0);
			e1[k] = ml_at(&mls[k], virt[k], i1);
		}
		for (uint32_t p = pt_lo; p < pt_hi; p++) {
			/* f(z, xs) = f(0, xs) + z (f(1, xs) - f(0, xs)); index 2 is the point at infinity: f(1) - f(0)
			 * (round_calculation.rs:186-232) */
			for (uint32_t k = 0; k < n_mls; k++) {
				if (p == 0) row[k] = e0[k];
				else if (p == 1) row[k] = e1[k];
				else if (p == 2) row[k] = ref_b128_add(e1[k], e0[k]);
				else row[k] = ref_b128_add(e0[k], ref_b128_mul(nontrivial_points[p - 3], ref_b128_add(e1[k], e0[k])));
			}
			size_t off = 0;
			for (uint32_t e = 0; e < n_evs; e++) {
				const uint32_t s = evs[e].eval_point_start, t = evs[e].eval_point_end;
				if (p >= s && p < t) {
					/* RegularSumcheckEvaluator / eq_ind Evaluator::process_subcube_at_eval_point
					 * (regular_sumcheck.rs:248-270, eq_ind.rs:676-704) */
					ref_b128 v = p == 2 ? ref_circuit_eval(evs[e].composition_at_infinity, evs[e].n_steps_inf, row)
					                    : ref_circuit_eval(evs[e].composition, evs[e].n_steps, row);
					if (evs[e].eq_ind) v = ref_b128_mul(v, evs[e].eq_ind[i]);
					out[off + (p - s)] = ref_b128_add(out[off + (p - s)], v);
				}
				off += t - s;
			}
		}
	}
	for (uint32_t k = 0; k < n_mls; k++) free(virt[k]);
	free(virt);
	free(e0);
	return rc;
}

int ref_hal_fold_multilinear(int order, uint32_t n_vars, const ref_hal_multilinear *ml, ref_b128 z,
                             const ref_b128 *tensor_query, uint32_t query_vars, ref_b128 *out, uint64_t out_cap, uint64_t *out_len)
{
	if (n_vars == 0) return 1;
	const uint64_t half = (uint64_t)1 << (n_vars - 1);
	if (ml->kind == REF_HAL_ML_TRANSPARENT) {
		/* switchover: the query already holds this round's challenge (prover_state.rs:158-171), so the partial
		 * evaluation has n_vars - 1 variables (sumcheck_folding.rs:58-112, 164-216) */
		if (out_cap < half || ml->n_vars_ml != n_vars - 1 + query_vars || query_vars == 0) return 1;
		ref_hal_multilinear t = *ml;
		const int rc = materialise(order, n_vars - 1, &t, tensor_query, query_vars, out);
		*out_len = half;
		return rc;
	}
	const uint64_t full = (uint64_t)1 << n_vars;
	const uint64_t len = ml->len < full ? ml->len : full;
	const ref_b128 *e = ml->evals;
	if (order == REF_ORDER_LOW_TO_HIGH) {
		/* fold_right_lerp (fold.rs:528-576): new length ceil(len / 2), an odd tail pairs with the suffix */
		const uint64_t n_out = (len + 1) / 2;
		if (out_cap < n_out) return 1;
		for (uint64_t i = 0; i < len / 2; i++)
			out[i] = ref_b128_add(e[2 * i], ref_b128_mul(z, ref_b128_add(e[2 * i + 1], e[2 * i])));
		if (len & 1)
			out[len / 2] = ref_b128_add(e[len - 1], ref_b128_mul(z, ref_b128_add(ml->suffix_eval, e[len - 1])));
		*out_len = n_out;
		return 0;
	}
	/* fold_left_lerp_inplace (fold.rs:648-696): prefix entries beyond the stored length equal the suffix */
	const uint64_t pivot = len > half ? len - half : 0;
	const uint64_t upper = len < half ? len : half;
	if (out_cap < upper) return 1;
	for (uint64_t i = 0; i < pivot; i++)
		out[i] = ref_b128_add(e[i], ref_b128_mul(z, ref_b128_add(e[half + i], e[i])));
	for (uint64_t i = pivot; i < upper; i++)
		out[i] = ref_b128_add(e[i], ref_b128_mul(z, ref_b128_add(ml->suffix_eval, e[i])));
	*out_len = upper;
	return 0;
}
