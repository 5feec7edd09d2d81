/*
 * oracle/ntt_ref.c -- TEST INFRASTRUCTURE ONLY (see ntt_ref.h).
 */
#include "ntt_ref.h"

#include <stdlib.h>
#include <string.h>

/* twiddle.rs:311-313  (e, c) -> e^2 + c*e */
static uint64_t subspace_map(uint64_t e, uint64_t c, int level)
{
	return ref_gf_square(e, level) ^ ref_gf_mul(c, e, level);
}

/* twiddle.rs:244-306 precompute_subspace_evals for BinarySubspace::with_dim(log_domain)
 * (basis beta_i = 1 << i, binary_subspace.rs:33-38, binary_field.rs:600-607) */
int ref_ntt_s_evals(int level, int log_domain, uint64_t *s_evals)
{
	if (log_domain < 1 || log_domain > (1 << level) || log_domain > REF_NTT_MAX_DIM || level > 6)
		return 1;
	int d = log_domain;
	uint64_t norm[REF_NTT_MAX_DIM];
	memset(s_evals, 0, sizeof(uint64_t) * REF_NTT_MAX_DIM * REF_NTT_MAX_DIM);
	norm[0] = 1;
	for (int b = 0; b < d - 1; b++)
		s_evals[b] = 1ull << (b + 1);
	for (int i = 1; i < d; i++) {
		const uint64_t *prev = &s_evals[(i - 1) * REF_NTT_MAX_DIM];
		uint64_t *cur = &s_evals[i * REF_NTT_MAX_DIM];
		uint64_t norm_prev = norm[i - 1];
		norm[i] = subspace_map(prev[0], norm_prev, level);
		for (int b = 0; b < d - 1 - i; b++)
			cur[b] = subspace_map(prev[b + 1], norm_prev, level);
	}
	for (int i = 0; i < d; i++) {
		uint64_t inv = ref_gf_invert(norm[i], level);
		uint64_t *cur = &s_evals[i * REF_NTT_MAX_DIM];
		for (int b = 0; b < d - 1 - i; b++)
			cur[b] = ref_gf_mul(cur[b], inv, level);
	}
	return 0;
}

/* twiddle.rs:141-143, 163-168: OnTheFlyTwiddleAccess::get with offset 0, log_n = d-1-layer */
uint64_t ref_ntt_twiddle(const uint64_t *s_evals, int log_domain, int layer, uint64_t index)
{
	const uint64_t *row = &s_evals[layer * REF_NTT_MAX_DIM];
	int n_bits = log_domain - 1 - layer;
	uint64_t t = 0;
	for (int b = 0; b < n_bits; b++)
		if ((index >> b) & 1)
			t ^= row[b];
	return t;
}

uint64_t ref_ntt_get_subspace_eval(const uint64_t *s_evals, int log_domain, int i, uint64_t j)
{
	return ref_ntt_twiddle(s_evals, log_domain, log_domain - i, j);
}

static ref_b128 load_elem(const void *data, int elem_level, size_t idx)
{
	ref_b128 r = {0, 0};
	size_t nb = (size_t)1 << (elem_level - 3);
	memcpy(&r, (const uint8_t *)data + idx * nb, nb);
	return r;
}
static void store_elem(void *data, int elem_level, size_t idx, ref_b128 v)
{
	size_t nb = (size_t)1 << (elem_level - 3);
	memcpy((uint8_t *)data + idx * nb, &v, nb);
}

static ref_b128 mul_tw(ref_b128 v, uint64_t t, int tw_level)
{
	ref_b128 s = {t, 0};
	return ref_b128_mul_subfield(v, s, tw_level);
}

static int check_args(int elem_level, int tw_level, int log_domain, int log_y, uint64_t coset, int coset_bits,
                      int skip_rounds)
{
	if (elem_level < 3 || elem_level > 7 || tw_level > elem_level || tw_level > 6)
		return 1;
	if (coset_bits < 64 && coset >= (1ull << coset_bits))
		return 1; /* Error::CosetIndexOutOfBounds */
	if (log_y + coset_bits > log_domain)
		return 1; /* Error::DomainTooSmall */
	if (skip_rounds > log_y)
		return 1;
	return 0;
}

/* tests/reference.rs:68-112 + batching :170-204 */
int ref_ntt_forward(void *data, int elem_level, int tw_level, const uint64_t *s_evals, int log_domain,
                    int log_x, int log_y, int log_z, uint64_t coset, int coset_bits, int skip_rounds)
{
	if (check_args(elem_level, tw_level, log_domain, log_y, coset, coset_bits, skip_rounds))
		return 1;
	int base = log_domain - (log_y + coset_bits); /* s_evals = &s_evals[base..] */
	int log_n = log_y;
	for (size_t x = 0; x < ((size_t)1 << log_x); x++)
		for (size_t z = 0; z < ((size_t)1 << log_z); z++) {
			size_t batch = x | z << (log_x + log_y);
			for (int i = log_n - skip_rounds - 1; i >= 0; i--) {
				/* (the butterflies of one layer are independent: blocks j across the host threads when there are many of them,
				 * otherwise the butterflies k inside a block -- the arithmetic and its order per element are unchanged) */
				const int many = (log_n - 1 - i) >= 6;
#pragma omp parallel for schedule(static) if (many && log_n >= 16)
				for (size_t j = 0; j < ((size_t)1 << (log_n - 1 - i)); j++) {
					uint64_t tw = ref_ntt_twiddle(s_evals, log_domain, base + i, coset << (log_n - 1 - i) | j);
#pragma omp parallel for schedule(static) if (!many && log_n >= 16)
					for (size_t k = 0; k < ((size_t)1 << i); k++) {
						size_t idx0 = j << (i + 1) | k;
						size_t idx1 = idx0 | (size_t)1 << i;
						size_t p0 = batch + (idx0 << log_x), p1 = batch + (idx1 << log_x);
						ref_b128 u = load_elem(data, elem_level, p0);
						ref_b128 v = load_elem(data, elem_level, p1);
						u = ref_b128_add(u, mul_tw(v, tw, tw_level));
						v = ref_b128_add(v, u);
						store_elem(data, elem_level, p0, u);
						store_elem(data, elem_level, p1, v);
					}
				}
			}
		}
	return 0;
}

/* tests/reference.rs:115-160 */
int ref_ntt_inverse(void *data, int elem_level, int tw_level, const uint64_t *s_evals, int log_domain,
                    int log_x, int log_y, int log_z, uint64_t coset, int coset_bits, int skip_rounds)
{
	if (check_args(elem_level, tw_level, log_domain, log_y, coset, coset_bits, skip_rounds))
		return 1;
	int base = log_domain - (log_y + coset_bits);
	int log_n = log_y;
	for (size_t x = 0; x < ((size_t)1 << log_x); x++)
		for (size_t z = 0; z < ((size_t)1 << log_z); z++) {
			size_t batch = x | z << (log_x + log_y);
			for (int i = 0; i < log_n - skip_rounds; i++) {
				const int many = (log_n - 1 - i) >= 6;
#pragma omp parallel for schedule(static) if (many && log_n >= 16)
				for (size_t j = 0; j < ((size_t)1 << (log_n - 1 - i)); j++) {
					uint64_t tw = ref_ntt_twiddle(s_evals, log_domain, base + i, coset << (log_n - 1 - i) | j);
#pragma omp parallel for schedule(static) if (!many && log_n >= 16)
					for (size_t k = 0; k < ((size_t)1 << i); k++) {
						size_t idx0 = j << (i + 1) | k;
						size_t idx1 = idx0 | (size_t)1 << i;
						size_t p0 = batch + (idx0 << log_x), p1 = batch + (idx1 << log_x);
						ref_b128 u = load_elem(data, elem_level, p0);
						ref_b128 v = load_elem(data, elem_level, p1);
						v = ref_b128_add(v, u);
						u = ref_b128_add(u, mul_tw(v, tw, tw_level));
						store_elem(data, elem_level, p0, u);
						store_elem(data, elem_level, p1, v);
					}
				}
			}
		}
	return 0;
}

/* crates/math/src/univariate.rs:255-261  x0 + (x1 - x0) * z */
static ref_b128 extrapolate_line_scalar(ref_b128 x0, ref_b128 x1, ref_b128 z)
{
	return ref_b128_add(x0, ref_b128_mul(ref_b128_add(x1, x0), z));
}

/* crates/compute/src/cpu/layer.rs:304-391 */
int ref_fri_fold(const uint64_t *s_evals, int tw_level, int log_domain, int log_len, int log_batch_size,
                 const ref_b128 *challenges, size_t n_challenges, const ref_b128 *data_in, size_t in_len,
                 ref_b128 *data_out, size_t out_len)
{
	if (in_len != ((size_t)1 << (log_len + log_batch_size)))
		return 1;
	if (n_challenges < (size_t)log_batch_size)
		return 1;
	if (n_challenges > (size_t)(log_batch_size + log_len))
		return 1;
	if (out_len != ((size_t)1 << (log_len - (n_challenges - log_batch_size))))
		return 1;
	const ref_b128 *interleave_ch = challenges;
	const ref_b128 *fold_ch = challenges + log_batch_size;
	size_t n_fold = n_challenges - log_batch_size;
	size_t chunk = (size_t)1 << n_challenges;
	ref_b128 *values = (ref_b128 *)malloc(sizeof(ref_b128) * chunk);
	for (size_t chunk_index = 0; chunk_index < out_len; chunk_index++) {
		memcpy(values, data_in + chunk_index * chunk, sizeof(ref_b128) * chunk);
		size_t cur = chunk;
		for (int c = 0; c < log_batch_size; c++) {
			size_t nn = cur / 2;
			for (size_t o = 0; o < nn; o++)
				values[o] = extrapolate_line_scalar(values[o * 2], values[o * 2 + 1], interleave_ch[c]);
			cur = nn;
		}
		int ll = log_len;
		int ls = (int)n_fold;
		for (size_t c = 0; c < n_fold; c++) {
			for (size_t off = 0; off < ((size_t)1 << (ls - 1)); off++) {
				uint64_t t = ref_ntt_get_subspace_eval(s_evals, log_domain, ll, (chunk_index << (ls - 1)) | off);
				ref_b128 u = values[off << 1], v = values[(off << 1) | 1];
				v = ref_b128_add(v, u);
				u = ref_b128_add(u, mul_tw(v, t, tw_level));
				values[off] = extrapolate_line_scalar(u, v, fold_ch[c]);
			}
			ll -= 1;
			ls -= 1;
		}
		data_out[chunk_index] = values[0];
	}
	free(values);
	return 0;
}

/* ntt/src/fri.rs:27-74 fold_interleaved_allocated, :96-174 fold_pair/fold_chunk,
 * :200-245 fold_interleaved_chunk (P = scalar packing, LOG_WIDTH 0);
 * tensor = MultilinearQuery::expand(interleave_challenges) = eq-indicator expansion
 * (crates/math/src/tensor_prod_eq_ind.rs:35-77). */
int ref_fold_interleaved(const uint64_t *s_evals, int tw_level, int log_domain, int log_len, int log_batch_size,
                         const ref_b128 *challenges, size_t n_challenges, const ref_b128 *codeword,
                         size_t in_len, ref_b128 *out, size_t out_len)
{
	if (in_len != ((size_t)1 << (log_len + log_batch_size)) || n_challenges < (size_t)log_batch_size)
		return 1;
	size_t n_fold = n_challenges - log_batch_size;
	if (n_fold > (size_t)log_len || out_len != ((size_t)1 << (log_len - n_fold)))
		return 1;
	size_t tlen = (size_t)1 << log_batch_size;
	ref_b128 *tensor = (ref_b128 *)calloc(tlen, sizeof(ref_b128));
	tensor[0] = ref_b128_one();
	for (int i = 0; i < log_batch_size; i++) {
		size_t half = (size_t)1 << i;
		for (size_t h = 0; h < half; h++) {
			ref_b128 prod = ref_b128_mul(tensor[h], challenges[i]);
			tensor[h] = ref_b128_add(tensor[h], prod);
			tensor[half + h] = prod;
		}
	}
	size_t fold_chunk_size = (size_t)1 << n_fold;
	size_t chunk_size = (size_t)1 << n_challenges;
	ref_b128 *scratch = (ref_b128 *)malloc(sizeof(ref_b128) * fold_chunk_size);
	const ref_b128 *fold_ch = challenges + log_batch_size;
	for (size_t ci = 0; ci < out_len; ci++) {
		const ref_b128 *vals = codeword + ci * chunk_size;
		for (size_t s = 0; s < fold_chunk_size; s++) {
			ref_b128 acc = ref_b128_zero();
			for (size_t t = 0; t < tlen; t++)
				acc = ref_b128_add(acc, ref_b128_mul(vals[s * tlen + t], tensor[t]));
			scratch[s] = acc;
		}
		/* fold_chunk */
		int ll = log_len;
		int ls = (int)n_fold;
		for (size_t c = 0; c < n_fold; c++) {
			for (size_t off = 0; off < ((size_t)1 << (ls - 1)); off++) {
				uint64_t t = ref_ntt_get_subspace_eval(s_evals, log_domain, ll, (ci << (ls - 1)) | off);
				ref_b128 u = scratch[off << 1], v = scratch[(off << 1) | 1];
				v = ref_b128_add(v, u);
				u = ref_b128_add(u, mul_tw(v, t, tw_level));
				scratch[off] = extrapolate_line_scalar(u, v, fold_ch[c]);
			}
			ll -= 1;
			ls -= 1;
		}
		out[ci] = scratch[0];
	}
	free(scratch);
	free(tensor);
	return 0;
}
