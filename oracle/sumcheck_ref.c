/*
 * oracle/sumcheck_ref.c -- TEST INFRASTRUCTURE ONLY (see sumcheck_ref.h).
 */
#include "sumcheck_ref.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

ref_b128 ref_evaluate_univariate(const ref_b128 *coeffs, size_t n, ref_b128 x)
{
	ref_b128 e = ref_b128_zero();
	for (size_t i = n; i-- > 0;)
		e = ref_b128_add(ref_b128_mul(e, x), coeffs[i]);
	return e;
}

ref_b128 ref_mle_evaluate(const ref_b128 *evals, unsigned n_vars, const ref_b128 *point)
{
	size_t n = (size_t)1 << n_vars;
	ref_b128 *t = (ref_b128 *)calloc(n, sizeof(ref_b128));
	t[0] = ref_b128_one();
	for (unsigned i = 0; i < n_vars; i++) {
		size_t half = (size_t)1 << i;
		for (size_t h = 0; h < half; h++) {
			ref_b128 p = ref_b128_mul(t[h], point[i]);
			t[h] = ref_b128_add(t[h], p);
			t[half + h] = p;
		}
	}
	ref_b128 acc = ref_b128_zero();
	for (size_t i = 0; i < n; i++)
		acc = ref_b128_add(acc, ref_b128_mul(evals[i], t[i]));
	free(t);
	return acc;
}

/* ---- round evals (bivariate_product.rs:303-408) ---- */
typedef struct {
	const ref_b128 *const *multilins;
	const uint32_t *comps;
	size_t n_comps;
	const ref_b128 *coeffs; /* alpha^c */
	size_t half, begin, end;
	ref_b128 y1, yinf;
} re_job;

static void *re_worker(void *p)
{
	re_job *j = (re_job *)p;
	ref_b128 y1 = ref_b128_zero(), yinf = ref_b128_zero();
	for (size_t c = 0; c < j->n_comps; c++) {
		const ref_b128 *a = j->multilins[j->comps[2 * c]];
		const ref_b128 *b = j->multilins[j->comps[2 * c + 1]];
		ref_b128 s1 = ref_b128_zero(), sinf = ref_b128_zero();
		for (size_t i = j->begin; i < j->end; i++) {
			ref_b128 a1 = a[j->half + i], b1 = b[j->half + i];
			s1 = ref_b128_add(s1, ref_b128_mul(a1, b1));
			sinf = ref_b128_add(sinf, ref_b128_mul(ref_b128_add(a[i], a1), ref_b128_add(b[i], b1)));
		}
		/* sum_composition_evals: *accumulator += ret * batch_coeff (cpu/layer.rs:512) */
		y1 = ref_b128_add(y1, ref_b128_mul(s1, j->coeffs[c]));
		yinf = ref_b128_add(yinf, ref_b128_mul(sinf, j->coeffs[c]));
	}
	j->y1 = y1;
	j->yinf = yinf;
	return NULL;
}

int ref_round_evals(const ref_b128 *const *multilins, size_t m, unsigned n_vars, const uint32_t *comps,
                    size_t n_comps, ref_b128 batch_coeff, ref_b128 out[2], int threads)
{
	if (n_vars == 0)
		return 1;
	for (size_t c = 0; c < 2 * n_comps; c++)
		if (comps[c] >= m)
			return 1;
	size_t half = (size_t)1 << (n_vars - 1);
	ref_b128 *coeffs = (ref_b128 *)malloc(sizeof(ref_b128) * (n_comps ? n_comps : 1));
	ref_b128 p = ref_b128_one();
	for (size_t c = 0; c < n_comps; c++) { /* binius_field::util::powers */
		coeffs[c] = p;
		p = ref_b128_mul(p, batch_coeff);
	}
	if (threads < 1)
		threads = 1;
	size_t n_jobs = (size_t)threads * 2; /* fast_compute/src/layer.rs:228-230: 2 * threads chunks */
	if (threads == 1)
		n_jobs = 1;
	if (n_jobs > half)
		n_jobs = half;
	re_job *jobs = (re_job *)calloc(n_jobs, sizeof(re_job));
	pthread_t *tids = (pthread_t *)calloc(n_jobs, sizeof(pthread_t));
	for (size_t t = 0; t < n_jobs; t++) {
		jobs[t].multilins = multilins;
		jobs[t].comps = comps;
		jobs[t].n_comps = n_comps;
		jobs[t].coeffs = coeffs;
		jobs[t].half = half;
		jobs[t].begin = half * t / n_jobs;
		jobs[t].end = half * (t + 1) / n_jobs;
	}
	if (n_jobs == 1) {
		re_worker(&jobs[0]);
	} else {
		for (size_t t = 0; t < n_jobs; t++)
			pthread_create(&tids[t], NULL, re_worker, &jobs[t]);
		for (size_t t = 0; t < n_jobs; t++)
			pthread_join(tids[t], NULL);
	}
	out[0] = ref_b128_zero();
	out[1] = ref_b128_zero();
	for (size_t t = 0; t < n_jobs; t++) {
		out[0] = ref_b128_add(out[0], jobs[t].y1);
		out[1] = ref_b128_add(out[1], jobs[t].yinf);
	}
	free(tids);
	free(jobs);
	free(coeffs);
	return 0;
}

/* ---- fold (bivariate_product.rs:168-232 -> extrapolate_line, cpu/layer.rs:393-408) ---- */
typedef struct {
	ref_b128 *evals;
	size_t half, begin, end;
	ref_b128 z;
} fold_job;

static void *fold_worker(void *p)
{
	fold_job *j = (fold_job *)p;
	for (size_t i = j->begin; i < j->end; i++) {
		ref_b128 x0 = j->evals[i], x1 = j->evals[j->half + i];
		j->evals[i] = ref_b128_add(x0, ref_b128_mul(ref_b128_add(x1, x0), j->z));
	}
	return NULL;
}

int ref_fold_high(ref_b128 *evals, unsigned n_vars, ref_b128 z, int threads)
{
	if (n_vars == 0)
		return 1;
	size_t half = (size_t)1 << (n_vars - 1);
	if (threads < 1)
		threads = 1;
	size_t n_jobs = threads == 1 ? 1 : (size_t)threads * 2;
	if (n_jobs > half)
		n_jobs = half;
	fold_job *jobs = (fold_job *)calloc(n_jobs, sizeof(fold_job));
	pthread_t *tids = (pthread_t *)calloc(n_jobs, sizeof(pthread_t));
	for (size_t t = 0; t < n_jobs; t++) {
		jobs[t].evals = evals;
		jobs[t].half = half;
		jobs[t].z = z;
		jobs[t].begin = half * t / n_jobs;
		jobs[t].end = half * (t + 1) / n_jobs;
	}
	if (n_jobs == 1) {
		fold_worker(&jobs[0]);
	} else {
		for (size_t t = 0; t < n_jobs; t++)
			pthread_create(&tids[t], NULL, fold_worker, &jobs[t]);
		for (size_t t = 0; t < n_jobs; t++)
			pthread_join(tids[t], NULL);
	}
	free(tids);
	free(jobs);
	return 0;
}

int ref_bivariate_sumcheck_prove(ref_b128 *const *multilins, size_t m, unsigned n_vars, const uint32_t *comps,
                                 size_t n_comps, const ref_b128 *sums, ref_b128 batch_coeff,
                                 const ref_b128 *challenges, ref_b128 *round_coeffs_out,
                                 ref_b128 *final_evals_out, int threads)
{
	/* PhaseState::InitialSums -> evaluate_univariate(sums, batch_coeff) (:150-156) */
	ref_b128 batched_sum = ref_evaluate_univariate(sums, n_comps, batch_coeff);
	for (unsigned round = 0; round < n_vars; round++) {
		unsigned rem = n_vars - round;
		ref_b128 ev[2];
		if (ref_round_evals((const ref_b128 *const *)multilins, m, rem, comps, n_comps, batch_coeff, ev, threads))
			return 1;
		/* calculate_round_coeffs_from_evals (:410-424) */
		ref_b128 y1 = ev[0], yinf = ev[1];
		ref_b128 c0 = ref_b128_add(batched_sum, y1);
		ref_b128 c2 = yinf;
		ref_b128 c1 = ref_b128_add(ref_b128_add(y1, c0), c2);
		ref_b128 *rc = &round_coeffs_out[3 * round];
		rc[0] = c0;
		rc[1] = c1;
		rc[2] = c2;
		/* fold (:168-232) */
		batched_sum = ref_evaluate_univariate(rc, 3, challenges[round]);
		for (size_t j = 0; j < m; j++)
			ref_fold_high(multilins[j], rem, challenges[round], threads);
	}
	for (size_t j = 0; j < m; j++)
		final_evals_out[j] = multilins[j][0];
	return 0;
}

/* ---- MLE-check prover (v3/bivariate_mlecheck.rs) ------------------------------------------- */

/* calculate_round_evals (:391-520): compositions are  multilin_i * multilin_j * eq_ind  where the
 * eq indicator chunk is the same for the evaluation at 1 and at infinity */
int ref_round_evals_eq(const ref_b128 *const *multilins, size_t m, unsigned n_vars, const ref_b128 *eq_ind,
                       const uint32_t *comps, size_t n_comps, ref_b128 batch_coeff, ref_b128 out[2])
{
	if (n_vars < 1) return 1;
	const size_t half = (size_t)1 << (n_vars - 1);
	ref_b128 acc1 = ref_b128_zero(), accinf = ref_b128_zero();
	ref_b128 coeff = ref_b128_one();
	for (size_t c = 0; c < n_comps; c++) {
		const uint32_t i = comps[2 * c], j = comps[2 * c + 1];
		if (i >= m || j >= m) return 1;
		const ref_b128 *a = multilins[i], *b = multilins[j];
		ref_b128 s1 = ref_b128_zero(), sinf = ref_b128_zero();
		for (size_t k = 0; k < half; k++) {
			ref_b128 p1 = ref_b128_mul(ref_b128_mul(a[half + k], b[half + k]), eq_ind[k]);
			ref_b128 pinf = ref_b128_mul(ref_b128_mul(ref_b128_add(a[k], a[half + k]), ref_b128_add(b[k], b[half + k])), eq_ind[k]);
			s1 = ref_b128_add(s1, p1);
			sinf = ref_b128_add(sinf, pinf);
		}
		acc1 = ref_b128_add(acc1, ref_b128_mul(s1, coeff));     /* *accumulator += ret * batch_coeff (cpu/layer.rs:512) */
		accinf = ref_b128_add(accinf, ref_b128_mul(sinf, coeff));
		coeff = ref_b128_mul(coeff, batch_coeff);                /* powers(batch_coeff) */
	}
	out[0] = acc1;
	out[1] = accinf;
	return 0;
}

int ref_bivariate_mlecheck_prove(ref_b128 *const *multilins, size_t m, unsigned n_vars, ref_b128 *eq_ind,
                                 const ref_b128 *eq_ind_challenges, const uint32_t *comps, size_t n_comps,
                                 const ref_b128 *sums, ref_b128 batch_coeff, const ref_b128 *challenges,
                                 ref_b128 *round_coeffs_out, ref_b128 *final_evals_out)
{
	const ref_b128 one = ref_b128_one();
	ref_b128 batched_sum = ref_evaluate_univariate(sums, n_comps, batch_coeff); /* InitialSums (:291) */
	ref_b128 prefix = one;                                                       /* eq_ind_prefix_eval */
	for (unsigned round = 0; round < n_vars; round++) {
		const unsigned rem = n_vars - round;
		ref_b128 ev[2];
		if (ref_round_evals_eq((const ref_b128 *const *)multilins, m, rem, eq_ind, comps, n_comps, batch_coeff, ev)) return 1;
		const ref_b128 alpha = eq_ind_challenges[rem - 1];
		/* calculate_round_coeffs_from_evals (:375-389): y_0 = (sum - y_1 alpha) / (1 - alpha) */
		const ref_b128 y1 = ev[0], yinf = ev[1];
		const ref_b128 y0 = ref_b128_mul(ref_b128_add(batched_sum, ref_b128_mul(y1, alpha)), ref_b128_invert(ref_b128_add(one, alpha)));
		ref_b128 prime[3];
		prime[0] = y0;
		prime[2] = yinf;
		prime[1] = ref_b128_add(ref_b128_add(y1, prime[0]), prime[2]);
		/* v(X) = v'(X) * eq(X, alpha) * prefix, eq(X, alpha) = (1 - alpha) + (2 alpha - 1) X; in
		 * characteristic 2: alpha.double() = 0, so the linear term is  -1 = 1   (:303-313) */
		const ref_b128 k0 = ref_b128_add(one, alpha), k1 = one;
		ref_b128 *rc = &round_coeffs_out[4 * round];
		for (int d = 0; d < 4; d++) {
			ref_b128 v = ref_b128_zero();
			if (d < 3) v = ref_b128_add(v, ref_b128_mul(prime[d], k0));
			if (d >= 1) v = ref_b128_add(v, ref_b128_mul(prime[d - 1], k1));
			rc[d] = ref_b128_mul(v, prefix);
		}
		/* fold (:320-346): the stored state is the PRIME polynomial (last_coeffs_or_sums = Coeffs(prime)) */
		const ref_b128 z = challenges[round];
		batched_sum = ref_evaluate_univariate(prime, 3, z);
		prefix = ref_b128_mul(prefix, ref_b128_add(ref_b128_add(alpha, z), one)); /* eq(alpha, z) = alpha + z + 1 */
		for (size_t j = 0; j < m; j++)
			ref_fold_high(multilins[j], rem, z, 1);
		if (rem - 1 != 0) {
			/* fold_eq_ind (:195-254): evals_0[i] += evals_1[i] over the halves of the 2^(rem-1) table */
			const size_t h = (size_t)1 << (rem - 2);
			for (size_t i = 0; i < h; i++)
				eq_ind[i] = ref_b128_add(eq_ind[i], eq_ind[h + i]);
		}
	}
	for (size_t j = 0; j < m; j++)
		final_evals_out[j] = multilins[j][0];
	final_evals_out[m] = prefix;
	return 0;
}

