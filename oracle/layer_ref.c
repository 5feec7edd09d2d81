/*
 * oracle/layer_ref.c -- TEST INFRASTRUCTURE ONLY (see layer_ref.h).
 * Restates crates/compute/src/cpu/layer.rs op by op; line references are to that file unless
 * another file is named.
 */
#include "layer_ref.h"

#include <stdlib.h>
#include <string.h>

static int is_pow2(size_t n) { return n && !(n & (n - 1)); }
static unsigned ilog2(size_t n)
{
	unsigned l = 0;
	while (n > 1) {
		n >>= 1;
		l++;
	}
	return l;
}

/* crates/math/src/arith_expr.rs:367-383 */
ref_b128 ref_circuit_eval(const ref_step *steps, size_t n_steps, const ref_b128 *query)
{
	if (n_steps == 0)
		return ref_b128_zero();
	ref_b128 *ev = (ref_b128 *)malloc(sizeof(ref_b128) * n_steps);
	for (size_t s = 0; s < n_steps; s++) {
		const ref_step *st = &steps[s];
		switch (st->kind) {
		case REF_STEP_ADD:
			ev[s] = ref_b128_add(ev[st->a], ev[st->b]);
			break;
		case REF_STEP_MUL:
			ev[s] = ref_b128_mul(ev[st->a], ev[st->b]);
			break;
		case REF_STEP_POW:
			ev[s] = ref_b128_pow(ev[st->a], st->b);
			break;
		case REF_STEP_CONST:
			ev[s] = st->cst;
			break;
		default:
			ev[s] = query[st->a];
			break;
		}
	}
	ref_b128 r = ev[n_steps - 1];
	free(ev);
	return r;
}

/* :393-408  x0 += (x1 - x0) * z */
int ref_extrapolate_line(ref_b128 *evals_0, const ref_b128 *evals_1, size_t n0, size_t n1, ref_b128 z)
{
	if (n0 != n1)
		return 1;
	for (size_t i = 0; i < n0; i++)
		evals_0[i] = ref_b128_add(evals_0[i], ref_b128_mul(ref_b128_add(evals_1[i], evals_0[i]), z));
	return 0;
}

/* :282-302 (CpuLayer: y += prod) ; crates/math/src/tensor_prod_eq_ind.rs:35-77 (y = prod) */
int ref_tensor_expand(ref_b128 *data, size_t data_len, size_t log_n, const ref_b128 *coords, size_t k, int assign)
{
	if (data_len != ((size_t)1 << (log_n + k)))
		return 1;
	for (size_t i = 0; i < k; i++) {
		size_t half = (size_t)1 << (log_n + i);
		for (size_t h = 0; h < half; h++) {
			ref_b128 prod = ref_b128_mul(data[h], coords[i]);
			data[h] = ref_b128_add(data[h], prod);
			data[half + h] = assign ? prod : ref_b128_add(data[half + h], prod);
		}
	}
	return 0;
}

/* the j-th T_iota limb of a (iter_bases order = least-significant limb first,
 * crates/field/src/binary_field.rs:633-649) */
static ref_b128 limb_of(const ref_b128 *a, size_t j, int iota)
{
	ref_b128 r = {0, 0};
	if (iota >= 7)
		return a[j];
	size_t per = (size_t)1 << (7 - iota);
	const ref_b128 *e = &a[j / per];
	unsigned w = 1u << iota;
	unsigned sh = (unsigned)(j % per) * w;
	uint64_t word = sh < 64 ? e->lo : e->hi;
	sh &= 63;
	uint64_t m = w >= 64 ? ~0ull : ((1ull << w) - 1);
	r.lo = (word >> sh) & m;
	return r;
}

static int valid_level(int l) { return l == 0 || (l >= 3 && l <= 7); } /* tower_macro.rs:9-15 */

/* :205-236 */
int ref_inner_product(const ref_b128 *a, size_t a_len, int tower_level, const ref_b128 *b, size_t b_len, ref_b128 *out)
{
	if (tower_level > 7 || tower_level < 0 || (a_len << (7 - tower_level)) != b_len)
		return 1;
	if (!valid_level(tower_level))
		return 1;
	ref_b128 acc = ref_b128_zero();
	for (size_t i = 0; i < b_len; i++)
		acc = ref_b128_add(acc, ref_b128_mul_subfield(b[i], limb_of(a, i, tower_level), tower_level));
	*out = acc;
	return 0;
}

/* :238-258, 574-621 */
int ref_fold_left(const ref_b128 *mat, size_t mat_len, int tower_level, const ref_b128 *vec, size_t vec_len,
                  ref_b128 *out, size_t out_len)
{
	if (tower_level > 7 || !valid_level(tower_level) || !is_pow2(mat_len) || !is_pow2(vec_len))
		return 1;
	size_t log_evals = ilog2(mat_len) + 7 - tower_level;
	size_t log_q = ilog2(vec_len);
	if (log_q > log_evals)
		return 1;
	size_t num_cols = (size_t)1 << log_q;
	size_t num_rows = (size_t)1 << (log_evals - log_q);
	if (out_len != num_rows)
		return 1;
	for (size_t i = 0; i < num_rows; i++) {
		ref_b128 acc = ref_b128_zero();
		for (size_t j = 0; j < num_cols; j++)
			acc = ref_b128_add(acc, ref_b128_mul_subfield(vec[j], limb_of(mat, j * num_rows + i, tower_level), tower_level));
		out[i] = acc;
	}
	return 0;
}

/* :260-280, 623-675 */
int ref_fold_right(const ref_b128 *mat, size_t mat_len, int tower_level, const ref_b128 *vec, size_t vec_len,
                   ref_b128 *out, size_t out_len)
{
	if (tower_level > 7 || !valid_level(tower_level) || !is_pow2(mat_len) || !is_pow2(vec_len))
		return 1;
	size_t log_evals = ilog2(mat_len) + 7 - tower_level;
	size_t log_q = ilog2(vec_len);
	if (log_q > log_evals)
		return 1;
	size_t num_rows = (size_t)1 << log_q;
	size_t num_cols = (size_t)1 << (log_evals - log_q);
	if (out_len != num_cols)
		return 1;
	for (size_t i = 0; i < num_cols; i++) {
		ref_b128 acc = ref_b128_zero();
		for (size_t j = 0; j < num_rows; j++)
			acc = ref_b128_add(acc, ref_b128_mul_subfield(vec[j], limb_of(mat, i * num_rows + j, tower_level), tower_level));
		out[i] = acc;
	}
	return 0;
}

/* :410-435 */
int ref_compute_composite(const ref_b128 *const *inputs, size_t n_rows, size_t row_len, ref_b128 *out,
                          size_t out_len, const ref_step *steps, size_t n_steps, size_t n_vars)
{
	if (row_len != out_len || n_vars != n_rows)
		return 1;
	ref_b128 *q = (ref_b128 *)malloc(sizeof(ref_b128) * (n_rows ? n_rows : 1));
	for (size_t i = 0; i < out_len; i++) {
		for (size_t j = 0; j < n_rows; j++)
			q[j] = inputs[j][i];
		out[i] = ref_circuit_eval(steps, n_steps, q);
	}
	free(q);
	return 0;
}

/* :437-484 */
int ref_pairwise_product_reduce(const ref_b128 *input, size_t n, ref_b128 *const *round_outputs,
                                const size_t *round_lens, size_t n_rounds)
{
	if (!is_pow2(n) || n < 2)
		return 1;
	size_t log_n = ilog2(n);
	if (n_rounds != log_n)
		return 1;
	for (size_t r = 0; r < n_rounds; r++)
		if (round_lens[r] != ((size_t)1 << (log_n - r - 1)))
			return 1;
	const ref_b128 *src = input;
	for (size_t r = 0; r < n_rounds; r++) {
		ref_b128 *dst = round_outputs[r];
		for (size_t i = 0; i < round_lens[r]; i++)
			dst[i] = ref_b128_mul(src[2 * i], src[2 * i + 1]);
		src = dst;
	}
	return 0;
}

/* :534-549 */
int ref_add_assign(ref_b128 *dst, const ref_b128 *src, size_t n)
{
	for (size_t i = 0; i < n; i++)
		dst[i] = ref_b128_add(dst[i], src[i]);
	return 0;
}

/* crates/compute/src/layer.rs:617-644 with Mem::ALIGNMENT == 1 */
int ref_log_chunks_range(const ref_memmap *maps, size_t n_maps, uint32_t *start, uint32_t *end)
{
	if (n_maps == 0)
		return 1;
	uint32_t s = 0, e = ~0u;
	for (size_t i = 0; i < n_maps; i++) {
		uint32_t hi;
		if (maps[i].kind == REF_MAP_LOCAL) {
			hi = maps[i].log_size;
		} else {
			uint32_t log_data = ilog2(maps[i].len);
			uint32_t lm = maps[i].log_min_chunk_size;
			if (lm > log_data)
				lm = log_data;
			hi = log_data - lm;
		}
		if (hi < e)
			e = hi;
	}
	*start = s;
	*end = e;
	return 0;
}

/* :92-153 map_kernel_mem + process_kernels_chunks, :168-203, :488-549 CpuKernelBuilder */
int ref_run_kernels(const ref_memmap *maps, size_t n_maps, const ref_kop *ops, size_t n_ops,
                    const uint32_t *ret_values, size_t n_ret, ref_b128 *out, int log_chunks)
{
	uint32_t s, e;
	if (ref_log_chunks_range(maps, n_maps, &s, &e))
		return 1;
	if (log_chunks < 0)
		log_chunks = (int)e; /* "For the reference implementation, use the smallest chunk size." */
	size_t n_chunks = (size_t)1 << log_chunks;

	ref_b128 **bufs = (ref_b128 **)calloc(n_maps, sizeof(ref_b128 *));
	size_t *lens = (size_t *)calloc(n_maps, sizeof(size_t));
	ref_b128 **locals = (ref_b128 **)calloc(n_maps, sizeof(ref_b128 *));
	size_t n_values = 0;
	for (size_t o = 0; o < n_ops; o++)
		if (ops[o].kind == REF_KOP_DECL_VALUE && ops[o].value + 1 > n_values)
			n_values = ops[o].value + 1;
	ref_b128 *values = (ref_b128 *)calloc(n_values ? n_values : 1, sizeof(ref_b128));
	for (size_t i = 0; i < n_ret; i++)
		out[i] = ref_b128_zero();

	for (size_t i = 0; i < n_maps; i++) {
		if (maps[i].kind == REF_MAP_LOCAL) {
			lens[i] = (size_t)1 << (maps[i].log_size - log_chunks);
			locals[i] = (ref_b128 *)malloc(sizeof(ref_b128) * lens[i]);
		} else {
			lens[i] = maps[i].len >> log_chunks;
		}
	}

	int rc = 0;
	for (size_t c = 0; c < n_chunks && !rc; c++) {
		for (size_t i = 0; i < n_maps; i++) {
			if (maps[i].kind == REF_MAP_LOCAL) {
				/* layer.rs:154-156: "a local scratchpad initialized with zeros" */
				memset(locals[i], 0, sizeof(ref_b128) * lens[i]);
				bufs[i] = locals[i];
			} else {
				bufs[i] = maps[i].data + c * lens[i];
			}
		}
		for (size_t o = 0; o < n_ops && !rc; o++) {
			const ref_kop *op = &ops[o];
			switch (op->kind) {
			case REF_KOP_DECL_VALUE:
				values[op->value] = op->scalar;
				break;
			case REF_KOP_SUM_COMPOSITION: {
				/* :499-514 */
				size_t row_len = op->n_rows ? op->rows[0].len : 0;
				ref_b128 sum = ref_b128_zero();
				ref_b128 *q = (ref_b128 *)malloc(sizeof(ref_b128) * (op->n_rows ? op->n_rows : 1));
				for (size_t i = 0; i < row_len; i++) {
					for (size_t j = 0; j < op->n_rows; j++)
						q[j] = bufs[op->rows[j].buf][op->rows[j].off + i];
					sum = ref_b128_add(sum, ref_circuit_eval(op->steps, op->n_steps, q));
				}
				free(q);
				values[op->value] = ref_b128_add(values[op->value], ref_b128_mul(sum, op->scalar));
				break;
			}
			case REF_KOP_ADD: {
				/* :516-532 */
				if (maps[op->dst.buf].kind == REF_MAP_CHUNKED) {
					rc = 1;
					break;
				}
				const ref_b128 *s1 = bufs[op->src1.buf] + op->src1.off;
				const ref_b128 *s2 = bufs[op->src2.buf] + op->src2.off;
				ref_b128 *d = bufs[op->dst.buf] + op->dst.off;
				for (size_t i = 0; i < op->dst.len; i++)
					d[i] = ref_b128_add(s1[i], s2[i]);
				break;
			}
			case REF_KOP_ADD_ASSIGN: {
				/* :534-549 */
				if (maps[op->dst.buf].kind == REF_MAP_CHUNKED) {
					rc = 1;
					break;
				}
				const ref_b128 *s1 = bufs[op->src1.buf] + op->src1.off;
				ref_b128 *d = bufs[op->dst.buf] + op->dst.off;
				for (size_t i = 0; i < op->dst.len; i++)
					d[i] = ref_b128_add(d[i], s1[i]);
				break;
			}
			default:
				rc = 1;
			}
		}
		/* :178-188 accumulate returned scalars with field addition */
		for (size_t i = 0; i < n_ret; i++)
			out[i] = ref_b128_add(out[i], values[ret_values[i]]);
	}

	for (size_t i = 0; i < n_maps; i++)
		free(locals[i]);
	free(locals);
	free(lens);
	free(bufs);
	free(values);
	return rc;
}
