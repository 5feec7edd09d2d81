/*
 * oracle/layer_ref.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's scalar `CpuLayer` (crates/compute/src/cpu/layer.rs),
 * op by op, over host arrays of ref_b128.  "Device memory" here is plain host memory, exactly
 * as in the reference's CpuLayer (cpu/layer.rs:678-693).  Used only by tests/, smoke() and
 * bench.py's cpu_baseline leg as the checker.
 */
#ifndef BINIUS_ORACLE_LAYER_REF_H
#define BINIUS_ORACLE_LAYER_REF_H

#include "gf2k_ref.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ArithCircuit (crates/math/src/arith_expr.rs:200-206, evaluate :367-383) ---- */
enum { REF_STEP_ADD = 0, REF_STEP_MUL = 1, REF_STEP_POW = 2, REF_STEP_CONST = 3, REF_STEP_VAR = 4 };
typedef struct {
	uint32_t kind;
	uint32_t a;     /* Add/Mul: left; Pow: base; Var: index */
	uint64_t b;     /* Add/Mul: right; Pow: exponent */
	ref_b128 cst;   /* Const */
} ref_step;

ref_b128 ref_circuit_eval(const ref_step *steps, size_t n_steps, const ref_b128 *query);

/* ---- ComputeLayerExecutor ops (crates/compute/src/cpu/layer.rs:205-485) ---- */
/* all return 0 on success, 1 on the reference's Error::InputValidation conditions */
int ref_extrapolate_line(ref_b128 *evals_0, const ref_b128 *evals_1, size_t n0, size_t n1, ref_b128 z);
/* assign != 0: y = prod (documented semantics, FastCpuLayer); assign == 0: y += prod (CpuLayer literal) */
int ref_tensor_expand(ref_b128 *data, size_t data_len, size_t log_n, const ref_b128 *coords, size_t k, int assign);
int ref_inner_product(const ref_b128 *a, size_t a_len, int tower_level, const ref_b128 *b, size_t b_len, ref_b128 *out);
int ref_fold_left(const ref_b128 *mat, size_t mat_len, int tower_level, const ref_b128 *vec, size_t vec_len,
                  ref_b128 *out, size_t out_len);
int ref_fold_right(const ref_b128 *mat, size_t mat_len, int tower_level, const ref_b128 *vec, size_t vec_len,
                   ref_b128 *out, size_t out_len);
int ref_compute_composite(const ref_b128 *const *inputs, size_t n_rows, size_t row_len, ref_b128 *out,
                          size_t out_len, const ref_step *steps, size_t n_steps, size_t n_vars);
int ref_pairwise_product_reduce(const ref_b128 *input, size_t n, ref_b128 *const *round_outputs,
                                const size_t *round_lens, size_t n_rounds);
int ref_add_assign(ref_b128 *dst, const ref_b128 *src, size_t n);

/* ---- accumulate_kernels / map_kernels (cpu/layer.rs:128-203, 488-549; layer.rs:595-677) ---- */
enum { REF_MAP_CHUNKED = 0, REF_MAP_CHUNKED_MUT = 1, REF_MAP_LOCAL = 2 };
typedef struct {
	uint32_t kind;
	uint32_t log_min_chunk_size;
	ref_b128 *data; /* Chunked / ChunkedMut */
	uint64_t len;
	uint32_t log_size; /* Local: total size over all chunks */
} ref_memmap;

/* a slice of kernel buffer `buf`, chunk-relative */
typedef struct {
	uint32_t buf;
	uint64_t off, len;
} ref_kslice;

enum { REF_KOP_DECL_VALUE = 0, REF_KOP_SUM_COMPOSITION = 1, REF_KOP_ADD = 2, REF_KOP_ADD_ASSIGN = 3 };
typedef struct {
	uint32_t kind;
	uint32_t value;          /* DECL_VALUE: id being declared; SUM_COMPOSITION: accumulator id */
	ref_b128 scalar;         /* DECL_VALUE: init; SUM_COMPOSITION: batch_coeff */
	const ref_step *steps;   /* SUM_COMPOSITION */
	uint32_t n_steps;
	uint32_t n_rows;
	const ref_kslice *rows;  /* SUM_COMPOSITION inputs */
	ref_kslice src1, src2, dst; /* ADD: dst = src1 + src2 ; ADD_ASSIGN: dst += src1 */
} ref_kop;

/* KernelMemMap::log_chunks_range (layer.rs:617-644), ALIGNMENT = 1. Returns 0 and sets
 * [*start, *end) or returns 1 for an empty mapping list. */
int ref_log_chunks_range(const ref_memmap *maps, size_t n_maps, uint32_t *start, uint32_t *end);

/* Runs the recorded kernel once per chunk with log_chunks chosen like CpuLayer (range.end, i.e.
 * the smallest chunks, cpu/layer.rs:141-142) unless force_log_chunks >= 0. The `ops` were
 * recorded for that log_chunks. Returned values `ret_values[i]` (ids) are XOR-accumulated over
 * chunks into out[i] (cpu/layer.rs:178-188). n_ret == 0 is map_kernels. */
int ref_run_kernels(const ref_memmap *maps, size_t n_maps, const ref_kop *ops, size_t n_ops,
                    const uint32_t *ret_values, size_t n_ret, ref_b128 *out, int log_chunks);

#ifdef __cplusplus
}
#endif
#endif
