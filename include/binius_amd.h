/*
 * binius_amd.h -- C ABI of the MI355X (gfx950) compute backend for the Binius prover hot path.
 *
 * Drop-in boundary: these entry points are what an implementation of the reference's HAL traits
 *   binius_compute::ComputeLayer          crates/compute/src/layer.rs:22
 *   binius_compute::ComputeLayerExecutor  crates/compute/src/layer.rs:100
 *   binius_compute::KernelExecutor        crates/compute/src/layer.rs:518
 *   binius_ntt::AdditiveNTT               crates/ntt/src/additive_ntt.rs:58
 * binds over FFI (INTEGRATION.md shows the Rust `extern "C"` block and the trait impl).
 * Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - F = BinaryField128b = one little-endian u128 = bn_f128 {lo, hi}
 *     (crates/field/src/binary_field.rs:747); element i of a slice is at byte offset 16*i.
 *   - Pointers named d_* are device pointers (16-byte aligned); h_* are host pointers.
 *     All lengths are in field elements unless stated otherwise.
 *   - Every function returns a bn_status.  Non-zero mirrors binius_compute::Error
 *     (crates/compute/src/layer.rs:706-716); bn_last_error() gives the message for the calling
 *     thread.  Precondition violations the reference reports as Error::InputValidation are
 *     reported as BN_ERR_INPUT_VALIDATION with nothing launched.
 *   - Ownership: the caller owns every buffer; the library owns its context (stream, scratch).
 *   - Ordering: work is enqueued on the context's HIP stream in call order, which gives the
 *     store-to-load ordering the executor contract asks for (layer.rs:92-95).  Functions that
 *     return scalars to the host synchronise the stream.
 */
#ifndef BINIUS_AMD_H
#define BINIUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
	uint64_t lo, hi;
} bn_f128;

typedef enum {
	BN_OK = 0,
	BN_ERR_INPUT_VALIDATION = 1, /* Error::InputValidation */
	BN_ERR_ALLOC = 2,            /* Error::Alloc(OutOfMemory) */
	BN_ERR_DEVICE = 3,           /* Error::DeviceError (hipError_t) */
	BN_ERR_CORE_LIB = 4          /* Error::CoreLibError */
} bn_status;

typedef struct bn_ctx bn_ctx;
typedef struct bn_expr bn_expr;

const char *bn_last_error(void);
/* "gfx950" build tag + ABI version, for the loader to check */
const char *bn_version(void);

/* ---- context = ComputeHolder (layer.rs:732-776; FastCpuLayerHolder::new(host, dev),
 *      crates/fast_compute/src/layer.rs:905).  arena_elems may be 0 (caller brings its own device
 *      memory, e.g. a torch allocation); otherwise one arena of arena_elems F is hipMalloc'ed and
 *      the host side bump-allocates from it (crates/compute/src/alloc.rs:31). */
int bn_ctx_create(int device, uint64_t arena_elems, bn_ctx **out);
int bn_ctx_destroy(bn_ctx *ctx);
int bn_arena_base(bn_ctx *ctx, void **d_base, uint64_t *elems);
/* Stream contract.  Work is enqueued on the context's stream in call order, with ONE exception: on the
 * context's OWN stream (the default) bn_extrapolate_line[_batch] and bn_copy_d2d may be deferred to the
 * next bn_* call (which either launches them first or fuses them into its own kernel) -- invisible to
 * every bn_* call, host read-back included.  Anything the caller enqueues on that stream itself must
 * come after bn_ctx_get_stream or bn_sync: both flush deferred work first.
 * bn_ctx_set_stream(s != NULL) runs the context on a caller-owned hipStream_t (e.g.
 * torch.cuda.current_stream().cuda_stream) and turns the deferral OFF (strict call order, no fold +
 * evaluation fusion) unless BN_LAZY_ON_SHARED_STREAM=1 accepts the rule above; NULL = back to an own
 * stream.  Every entry point makes the context's device current on the calling thread. */
int bn_ctx_set_stream(bn_ctx *ctx, void *hip_stream);
int bn_sync(bn_ctx *ctx);
/* the hipStream_t the context enqueues on, after flushing deferred work (so a caller can put an RCCL
 * collective in the same order) */
int bn_ctx_get_stream(bn_ctx *ctx, void **hip_stream);

/* ---- ComputeLayer (layer.rs:36-87) ---- */
int bn_copy_h2d(bn_ctx *ctx, const bn_f128 *h_src, uint64_t src_len, void *d_dst, uint64_t dst_len);
int bn_copy_d2h(bn_ctx *ctx, const void *d_src, uint64_t src_len, bn_f128 *h_dst, uint64_t dst_len);
int bn_copy_d2d(bn_ctx *ctx, const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len);
int bn_fill(bn_ctx *ctx, void *d_dst, uint64_t n, const bn_f128 *value);

/* compile_expr (layer.rs:57): ArithCircuitStep list, crates/math/src/arith_expr.rs:200-206 */
enum { BN_STEP_ADD = 0, BN_STEP_MUL = 1, BN_STEP_POW = 2, BN_STEP_CONST = 3, BN_STEP_VAR = 4 };
typedef struct {
	uint32_t kind;
	uint32_t a;   /* Add/Mul: left step; Pow: base step; Var: variable index */
	uint64_t b;   /* Add/Mul: right step; Pow: exponent */
	bn_f128 cst;  /* Const */
} bn_step;
int bn_expr_compile(bn_ctx *ctx, const bn_step *steps, uint64_t n_steps, bn_expr **out);
int bn_expr_free(bn_expr *expr);
int bn_expr_n_vars(const bn_expr *expr, uint32_t *n_vars);

/* ---- ComputeLayerExecutor ---- */
/* extrapolate_line (layer.rs:421): evals_0[i] += (evals_1[i] - evals_0[i]) * z */
int bn_extrapolate_line(bn_ctx *ctx, void *d_evals_0, uint64_t n0, const void *d_evals_1, uint64_t n1,
                        const bn_f128 *z);
/* `count` extrapolate_line calls with the same z and length issued inside one executor `map` scope
 * (ComputeLayerExecutor::map, layer.rs:126; the fold of all multilinears of a round,
 * v3/bivariate_product.rs:217-228) as one call.  count <= BN_FOLD_CALL_MAX per call: a BivariateSumcheckProver of the PCS
 * prover holds every committed multilinear of one size and its transparents (piop/prove.rs:262-287: >= 100 for keccak), and
 * the backend defers the whole fold as one batch whose arrays the next round evaluation's launch folds on the way. */
#define BN_FOLD_CALL_MAX 256
int bn_extrapolate_line_batch(bn_ctx *ctx, void *const *d_evals_0, const void *const *d_evals_1, uint32_t count, uint64_t n,
                              const bn_f128 *z);
/* Extension (not a trait method): the same batch fold, after which the UPPER half of every folded array i with bit i of
 * scale_mask set is multiplied by *hi_scale (n even).  Deferred and fused with the next round evaluation exactly like
 * bn_extrapolate_line_batch.  It lets a prover keep a multilinear pre-multiplied by the equality indicator across
 * rounds (the weighted MLE-check prover of binius_amd/host/sumcheck.hpp): when the variable that
 * splits the folded array becomes the round variable, its two halves carry the factors (1 - zeta) and zeta of that
 * variable's indicator coordinate; multiplying the upper half by (1 - zeta) / zeta levels them. */
int bn_extrapolate_line_batch_scaled(bn_ctx *ctx, void *const *d_evals_0, const void *const *d_evals_1, uint32_t count, uint64_t n,
                                     const bn_f128 *z, uint32_t scale_mask, const bn_f128 *hi_scale);
/* tensor_expand (layer.rs:291): data[..2^(log_n+k)] = data[..2^log_n] (x) (1-r_0,r_0) (x) ...
 * The upper half of every pass is OVERWRITTEN (y = prod), as in FastCpuLayer and
 * math/src/tensor_prod_eq_ind.rs:35-77.  The scalar CpuLayer ACCUMULATES into it instead (*y += prod,
 * cpu/layer.rs:298); the two agree whenever data[2^log_n ..] is zero on entry, which is what every
 * caller in crates/core guarantees (they allocate and zero-fill first).  A caller that relies on the
 * accumulate form with a non-zero tail must add the old tail itself. */
int bn_tensor_expand(bn_ctx *ctx, void *d_data, uint64_t data_len, uint32_t log_n, const bn_f128 *h_coords,
                     uint32_t k);
/* inner_product (layer.rs:263): a is a SubfieldSlice{slice, tower_level} (memory.rs:257-281) */
int bn_inner_product(bn_ctx *ctx, const void *d_a, uint64_t a_len, uint32_t tower_level, const void *d_b,
                     uint64_t b_len, bn_f128 *h_out);
/* fold_left / fold_right (layer.rs:321, 351) */
int bn_fold_left(bn_ctx *ctx, const void *d_mat, uint64_t mat_len, uint32_t tower_level, const void *d_vec,
                 uint64_t vec_len, void *d_out, uint64_t out_len);
int bn_fold_right(bn_ctx *ctx, const void *d_mat, uint64_t mat_len, uint32_t tower_level, const void *d_vec,
                  uint64_t vec_len, void *d_out, uint64_t out_len);
/* fri_fold (layer.rs:389).  The NTT object is passed as its on-the-fly twiddle basis:
 * h_s_evals[i*BN_NTT_MAX_DIM + b] = s_evals[i][b] of OnTheFlyTwiddleAccess
 * (crates/ntt/src/twiddle.rs:93-124), elements of T_tw_level in the low bits of a u64. */
#define BN_NTT_MAX_DIM 64
int bn_fri_fold(bn_ctx *ctx, const uint64_t *h_s_evals, uint32_t tw_level, uint32_t log_domain, uint32_t log_len,
                uint32_t log_batch_size, const bn_f128 *h_challenges, uint32_t n_challenges, const void *d_in,
                uint64_t in_len, void *d_out, uint64_t out_len);
/* compute_composite (layer.rs:459) */
int bn_compute_composite(bn_ctx *ctx, const void *const *d_rows, uint32_t n_rows, uint64_t row_len, void *d_out,
                         uint64_t out_len, const bn_expr *expr);
/* pairwise_product_reduce (layer.rs:505) */
int bn_pairwise_product_reduce(bn_ctx *ctx, const void *d_in, uint64_t n, void *const *d_round_outs,
                               const uint64_t *round_lens, uint32_t n_rounds);

/* ---- accumulate_kernels / map_kernels (layer.rs:183, 236) + KernelExecutor (layer.rs:518-590).
 * The kernel-spec closure cannot cross an FFI: the host shim runs it ONCE against a recording
 * KernelExecutor (the trait allows re-invocation, layer.rs:171-177) and passes the recorded op
 * list here.  Slices are chunk-relative views of the mapped buffers. */
enum { BN_MAP_CHUNKED = 0, BN_MAP_CHUNKED_MUT = 1, BN_MAP_LOCAL = 2 }; /* KernelMemMap, layer.rs:595-612 */
typedef struct {
	uint32_t kind;
	uint32_t log_min_chunk_size;
	void *d_data;     /* Chunked / ChunkedMut */
	uint64_t len;
	uint32_t log_size; /* Local: total size over all chunks */
} bn_memmap;

typedef struct {
	uint32_t buf;       /* index into the memmap list */
	uint64_t off, len;  /* within the chunk */
} bn_kslice;

enum { BN_KOP_DECL_VALUE = 0, BN_KOP_SUM_COMPOSITION = 1, BN_KOP_ADD = 2, BN_KOP_ADD_ASSIGN = 3 };
typedef struct {
	uint32_t kind;
	uint32_t value;          /* DECL_VALUE: id declared; SUM_COMPOSITION: accumulator id */
	bn_f128 scalar;          /* DECL_VALUE: init; SUM_COMPOSITION: batch_coeff */
	const bn_expr *expr;     /* SUM_COMPOSITION */
	uint32_t n_rows;
	const bn_kslice *rows;   /* SUM_COMPOSITION inputs */
	bn_kslice src1, src2, dst; /* ADD: dst = src1 + src2; ADD_ASSIGN: dst += src1 */
} bn_kop;

/* KernelMemMap::log_chunks_range (layer.rs:617-644), ComputeMemory::ALIGNMENT = 1. Host only. */
int bn_log_chunks_range(const bn_memmap *maps, uint32_t n_maps, uint32_t *start, uint32_t *end);
/* log_chunks this backend wants the closure recorded with (host only; deterministic). */
int bn_pick_log_chunks(const bn_memmap *maps, uint32_t n_maps, uint32_t *log_chunks);
/* Launch.  ops were recorded for `log_chunks`.  n_ret == 0 is map_kernels.  The n_ret returned
 * values (ids in ret_values) are XOR-accumulated over chunks (cpu/layer.rs:178-188) and written
 * to h_out (stream is synchronised) and/or left in d_out (n_ret elements, no sync) -- the latter
 * lets a multi-GPU caller feed them straight into an RCCL collective. */
int bn_kernel_launch(bn_ctx *ctx, const bn_memmap *maps, uint32_t n_maps, const bn_kop *ops, uint32_t n_ops,
                     const uint32_t *ret_values, uint32_t n_ret, uint32_t log_chunks, bn_f128 *h_out, void *d_out);

/* ---- AdditiveNTT (crates/ntt/src/additive_ntt.rs:102, 128): data is 2^(log_x+log_y+log_z)
 * elements of T_elem_level (elem_level 5 = BinaryField32b ... 7 = BinaryField128b), transform along
 * y; twiddles in T_tw_level given as the on-the-fly basis (see bn_fri_fold). */
/* ---- the OLD hardware abstraction layer: binius_hal::ComputationBackend (crates/hal/src/backend.rs:35-84), which the
 * v2 provers (zerocheck, evalcheck, RegularSumcheckProver) still run on, for device-resident multilinears.
 *   tensor_product_full_query   = bn_fill(1 element) + bn_tensor_expand            (cpu.rs:36-41)
 *   evaluate_partial_high       = bn_fold_left, evaluate_partial_low = bn_fold_right (multilinear_extension.rs:253-341)
 *   sumcheck_compute_round_evals = bn_hal_round_evals                               (sumcheck_round_calculation.rs:45-330)
 *   sumcheck_fold_multilinears   = bn_hal_fold_multilinear per multilinear           (sumcheck_folding.rs:16-237)
 * The trait takes its evaluators as trait objects; what crosses the boundary instead is their content: the
 * composition and its leading term as compiled circuits, the range of evaluation point indices, and the optional
 * equality-indicator table (RegularSumcheckEvaluator, regular_sumcheck.rs:219-282; eq_ind Evaluator, eq_ind.rs:646-735).
 * The switchover bookkeeping (switchover_round countdown, Transparent -> Folded) stays with the host-side state, as
 * in the reference (sumcheck_multilinear.rs:8-76).  Constant evaluation suffixes of EVALUATORS are not supported
 * (const_eval_suffix must be 0); constant suffixes of Folded multilinears are. */
enum { BN_ORDER_LOW_TO_HIGH = 0, BN_ORDER_HIGH_TO_LOW = 1 };  /* binius_math::EvaluationOrder */
enum { BN_HAL_ML_FOLDED = 0, BN_HAL_ML_TRANSPARENT = 1 };     /* SumcheckMultilinear */
typedef struct {
	uint32_t kind;
	uint32_t tower_level;  /* TRANSPARENT: level of the packed subfield values (0, 3..7) */
	const void *d_evals;   /* FOLDED: `len` evaluations; TRANSPARENT: the 2^n_vars_ml subfield values, packed into F */
	uint64_t len;          /* FOLDED: stored evaluations (the rest of the 2^n_vars cube equals suffix_eval); TRANSPARENT: F elements */
	bn_f128 suffix_eval;   /* FOLDED */
	uint32_t n_vars_ml;    /* TRANSPARENT: variables of the multilinear = n_vars + query_vars */
} bn_hal_multilinear;
typedef struct {
	const bn_expr *composition;
	const bn_expr *composition_at_infinity;     /* ArithCircuit::leading_term */
	uint32_t eval_point_start, eval_point_end;  /* SumcheckEvaluator::eval_point_indices: 0 -> X=0, 1 -> X=1, 2 -> infinity, 3+k -> point k */
	const void *d_eq_ind;                       /* NULL, or 2^(n_vars-1) factors (eq_ind partial evaluations) */
} bn_hal_evaluator;
/* h_out: evaluator by evaluator, one value per evaluation point index of its range.  d_tensor_query: the query
 * EXPANSION (2^query_vars elements) that TRANSPARENT multilinears are partially evaluated at (low variables for
 * BN_ORDER_LOW_TO_HIGH, high variables for BN_ORDER_HIGH_TO_LOW); NULL when query_vars == 0. */
int bn_hal_round_evals(bn_ctx *ctx, uint32_t order, uint32_t n_vars, const void *d_tensor_query, uint32_t query_vars,
                       const bn_hal_multilinear *mls, uint32_t n_mls, const bn_hal_evaluator *evaluators, uint32_t n_evaluators,
                       const bn_f128 *h_nontrivial_points, uint32_t n_points, bn_f128 *h_out);
/* FOLDED: single-variable lerp fold in the given order (d_out may be d_evals for BN_ORDER_HIGH_TO_LOW, must not overlap it
 * for BN_ORDER_LOW_TO_HIGH).  TRANSPARENT, at its switchover round: partial evaluation at the query, which already holds
 * this round's challenge (n_vars_ml = n_vars - 1 + query_vars).  *out_len evaluations are written. */
int bn_hal_fold_multilinear(bn_ctx *ctx, uint32_t order, uint32_t n_vars, const bn_hal_multilinear *ml, const bn_f128 *challenge,
                            const void *d_tensor_query, uint32_t query_vars, void *d_out, uint64_t out_cap, uint64_t *out_len);

int bn_ntt_forward(bn_ctx *ctx, void *d_data, uint32_t elem_level, uint32_t tw_level, const uint64_t *h_s_evals,
                   uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset,
                   uint32_t coset_bits, uint32_t skip_rounds);
int bn_ntt_inverse(bn_ctx *ctx, void *d_data, uint32_t elem_level, uint32_t tw_level, const uint64_t *h_s_evals,
                   uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset,
                   uint32_t coset_bits, uint32_t skip_rounds);
/* OnTheFlyTwiddleAccess::generate for the canonical subspace (twiddle.rs:107-124, 244-306). Host only. */
int bn_ntt_s_evals(uint32_t tw_level, uint32_t log_domain, uint64_t *h_s_evals);

/* ---- host-only scalar helpers for the O(1)-per-round protocol scalars the host keeps
 * (evaluate_univariate of the round polynomial, powers of the batch coefficient:
 * crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:150-156, 339-341).  Same arithmetic
 * as the kernels (gf128.hpp); never used on hypercube-sized data. */
int bn_scalar_mul(const bn_f128 *a, const bn_f128 *b, bn_f128 *out);
int bn_scalar_invert(const bn_f128 *a, bn_f128 *out);

/* ---- measurement plumbing: time the launches enqueued between begin and end with hipEvents on
 * the context's stream.  Not part of the reference interface. */
int bn_timer_begin(bn_ctx *ctx);
int bn_timer_end_ms(bn_ctx *ctx, float *ms);
/* 32 elements of fine-grained pinned host memory, readable by kernels through *d_ptr: a handful of
 * elements can be handed to the device without an upload.  Not part of the reference interface. */
int bn_host_scratch(bn_ctx *ctx, void **h_ptr, void **d_ptr, uint64_t *elems);
/* NUMA node of the host the device hangs off (its PCI function's numa_node in sysfs), -1 if the platform does not say.
 * A small round is a host -> device -> host round trip (the armed kernels of csrc/arm.hpp poll pinned host memory and
 * answer into it): from a core of the other socket every such trip crosses the socket interconnect as well -- measured
 * 15.0 against 17.1 us per two-round launch on a 2-socket host -- so the thread that drives a context should run on this
 * node.  The library never changes an affinity itself.  Not part of the reference interface. */
int bn_device_numa_node(int device, int *node);

/* out[i] = XOR over g < n_groups of d_vals[g * group_len + i], i < group_len <= 64, returned to the
 * host.  Not part of the reference interface: the combine step behind the per-round all_gather of
 * the multi-GPU prover (RCCL has no XOR reduction). */
int bn_xor_reduce(bn_ctx *ctx, const void *d_vals, uint32_t n_groups, uint32_t group_len, bn_f128 *h_out);

/* ---- Cross-rank reduction of round evaluations inside the kernel's finalize step (one process
 * per GPU, the hypercube sharded on the last-bound variables; the only exchange of a sumcheck round is one partial
 * (y_1, y_inf) per rank -- crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:303-408 is a sum over hypercube
 * points, so the shards' sums add).  Not part of the reference interface.
 *   bn_peer_create   allocates this rank's mailbox (BN_PEER_MAILBOX_BYTES of fine-grained device memory) and returns
 *                    its hipIpc handle (BN_PEER_HANDLE_BYTES) for the caller to hand to the other ranks of the node
 *   bn_peer_connect  maps every rank's mailbox (handles[world][BN_PEER_HANDLE_BYTES], rank-major; the own entry is
 *                    ignored) -- same-device peers on a one-GPU box, xGMI peers on a node
 *   bn_peer_set_active(1)  from now on every bn_kernel_launch that returns values to the host (h_out)
 *                    returns the XOR over all ranks of the values it would have returned alone: the finalizing workgroup
 *                    stores its values into every peer's mailbox, waits for the world's, XORs (csrc/finalize.hpp
 *                    peer_exchange).  All ranks must issue the same sequence of such launches.  Values declared with a
 *                    non-zero initial value are XORed in once per rank.  (0): back to local results (e.g. for the residual
 *                    rounds every rank runs identically).
 *   bn_peer_stats    stats[0] = reduced launches so far */
enum { BN_PEER_HANDLE_BYTES = 64, BN_PEER_MAX_WORLD = 16, BN_PEER_MAILBOX_BYTES = 2 * 16 * 24 * 8 };
int bn_peer_create(bn_ctx *ctx, uint32_t world, uint32_t rank, uint8_t *handle_out /*[BN_PEER_HANDLE_BYTES]*/);
int bn_peer_connect(bn_ctx *ctx, const uint8_t *handles /*[world][BN_PEER_HANDLE_BYTES]*/);
int bn_peer_set_active(bn_ctx *ctx, int on);
int bn_peer_stats(bn_ctx *ctx, uint64_t *stats /*[2]*/);
/* The host tail under a peer exchange (sharded sumcheck): once the arrays of a shard are down to <= 2^12 (2^8) elements the library
 * takes the remaining rounds onto the host (BN_ARM_HT_*), where the ranks' partial sums cannot meet on the devices any more.
 * A caller that exchanges the partials of those rounds ITSELF (host shared memory: binius_amd/host/host_capi.cpp) says so with
 * bn_host_tail_allow_peer(ctx, 1); after every reduced launch it asks bn_host_tail_active, and from the launch that answers 1 on
 * it switches the peer exchange off (bn_peer_set_active(ctx, 0): no flush while the tail is active) and combines what
 * bn_kernel_launch returns -- this rank's LOCAL sums -- across the ranks.  Without the permission a context with an active peer
 * exchange never starts a host tail. */
int bn_host_tail_allow_peer(bn_ctx *ctx, int on);
int bn_host_tail_active(bn_ctx *ctx, int *active);
int bn_peer_destroy(bn_ctx *ctx);

/* ---- Merkle commitment of a BinaryField128b vector with Groestl-256.
 * BinaryMerkleTreeProver::commit (crates/core/src/merkle_tree/prover.rs:47-62) =
 * binary_merkle_tree::build (binary_merkle_tree.rs:27-101) with H = Groestl256
 * (crates/hash/src/groestl/digest.rs:30-87) and C = Groestl256ByteCompression
 * (crates/hash/src/groestl/compression.rs:21-36), which the FRI prover runs on the host copy of every
 * folded codeword (crates/core/src/protocols/fri/prove.rs:395-420).
 * Leaf i = Groestl-256 of the canonical serialization (16 little-endian bytes per element,
 * crates/utils/src/serialization.rs:94-104) of elements [i * batch_size, (i + 1) * batch_size).
 * d_nodes receives the flattened tree, (2 * n_leaves - 1) digests of 32 bytes = 2 * (2 * n_leaves - 1)
 * arena elements, leaves first, root last (binary_merkle_tree.rs:22-25).  No synchronisation.
 * Errors as the reference: n_elems % batch_size != 0 -> "IncorrectBatchSize", leaf count not a power
 * of two -> "PowerOfTwoLengthRequired" (both BN_ERR_INPUT_VALIDATION). */
int bn_merkle_build(bn_ctx *ctx, const void *d_elems, uint64_t n_elems, uint64_t batch_size, void *d_nodes);
/* hash_interleaved alone (binary_merkle_tree.rs:175-211): d_digests[i] = Groestl-256(batch i), 32 B each. */
int bn_groestl256_leaves(bn_ctx *ctx, const void *d_elems, uint64_t n_elems, uint64_t batch_size, void *d_digests);
/* compress_layer alone (binary_merkle_tree.rs:158-168): d_next[i] = C(d_prev[2i], d_prev[2i+1]), i < n_out. */
int bn_groestl256_compress_layer(bn_ctx *ctx, const void *d_prev, uint64_t n_out, void *d_next);

/* Openings of a committed vector: h_out[i * item_elems + e] = d_src[h_offsets[i] + e] (offsets and
 * lengths in 16-byte elements), e < item_elems, i < n_items, in one kernel and one synchronisation.
 * Not part of the reference interface: the reference reads branches and cosets out of host copies
 * (binary_merkle_tree.rs:121-141, fri/prove.rs:631-661); here the tree and the codewords stay on the
 * device and only what a query opens is read back. */
int bn_gather_d2h(bn_ctx *ctx, const void *d_src, const uint64_t *h_offsets, uint64_t n_items, uint64_t item_elems, bn_f128 *h_out);

/* Per-kernel-class timing without perturbing the stream: while profiling is on, every launch of a
 * hot kernel is bracketed by two hipEvents recorded on the context's stream (no synchronisation);
 * bn_prof_end synchronises once and sums the elapsed times per class. */
enum { BN_PROF_ROUND_EVAL = 0, BN_PROF_FOLD = 1, BN_PROF_TENSOR_EXPAND = 2, BN_PROF_NTT = 3, BN_PROF_OTHER = 4, BN_PROF_FOLD_EVAL = 5, BN_PROF_TAIL = 6, BN_PROF_FOLD_EVAL_SMALL = 7, BN_PROF_FOLD_EVAL_MFMA = 8, BN_PROF_ROUND_EVAL_MFMA = 9, BN_PROF_FOLD_EVAL8 = 10, BN_PROF_N = 11 };
int bn_prof_begin(bn_ctx *ctx);
int bn_prof_end(bn_ctx *ctx, double *ms_by_class /*[BN_PROF_N]*/, uint64_t *launches_by_class /*[BN_PROF_N]*/);

/* Armed rounds (binius_amd/csrc/arm.hpp): behind the fused fold + evaluation kernel of a small sumcheck round the
 * dispatcher enqueues the kernel of the NEXT round, which waits on the device for its challenge; the caller's next
 * extrapolate_line + accumulate_kernels pair then costs a write to pinned memory instead of a launch.  Purely an
 * execution detail of bn_extrapolate_line_batch + bn_kernel_launch (the results are those of the unarmed path; BN_ARM=0
 * turns it off).  Counters since context creation: rounds served by an armed kernel, armed kernels that were cancelled
 * because the next call was something else, armed kernels that gave up waiting; host nanoseconds between handing over
 * a challenge and seeing the round's result, the part of that spent enqueueing the next armed kernel, and the time from
 * the entry of bn_kernel_launch to handing the challenge over (validation + recognising the round).
 * BN_ARM_HT_*: the host tail -- once the arrays of a bivariate sumcheck are down to <= 2^12 (2^8) elements the two-round kernel
 * hands them to the host and the remaining evaluations and folds are host arithmetic (PCLMULQDQ in an isomorphic power
 * basis), the device catching up with one launch; instances taken over, evaluations answered, catch-up launches, and (not a
 * counter) the largest array the host takes over: 2^12 elements when the host has VPCLMULQDQ (four products per instruction),
 * else 2^8; 0 when the host tail is off.  BN_HOST_TAIL=0 turns it off.  Like arming, an execution detail of the unchanged call
 * sequence. */
enum { BN_ARM_HITS = 0, BN_ARM_CANCELS = 1, BN_ARM_EXPIRED = 2, BN_ARM_NS_WAIT = 3, BN_ARM_NS_LAUNCH = 4, BN_ARM_NS_PARSE = 5, BN_ARM_HOSTED = 6, BN_ARM_TWO_ROUND = 7, BN_ARM_SHADOW_CREATED = 8, BN_ARM_SHADOW_ROUNDS = 9, BN_ARM_SHADOW_DROPPED = 10, BN_ARM_HT_STARTED = 11, BN_ARM_HT_ROUNDS = 12, BN_ARM_HT_FLUSHED = 13, BN_ARM_HT_MAX = 14, BN_ARM_N = 15 };
int bn_arm_counters(bn_ctx *ctx, uint64_t *counters /*[BN_ARM_N]*/);
/* Claim groups (not part of the reference interface: counters of how the backend ran the call shape of piop::prove --
 * k product claims over m multilinears per BivariateSumcheckProver, several provers front-loaded on one layer,
 * core/src/piop/prove.rs:271-287, protocols/sumcheck/prove/front_loaded.rs:122-155).  LAUNCHES: launches of the group kernel
 * (each answers one execute() and computes ahead for the other waiting provers); JOBS_FUSED / JOBS_EVAL: claims evaluated
 * together with the fold of their two arrays / on arrays folded by a plain launch (shared arrays, round 0); PREFOLDS: those
 * plain fold launches; SPEC_JOBS: claims of OTHER provers carried by a launch; SPEC_HITS: execute() calls answered from sums
 * computed ahead, without a launch; EVALS: execute() calls answered on this path; FLUSHED_FOLDS: deferred fold batches that a
 * foreign call forced out as plain launches. */
/* HOSTED_*: provers whose arrays were small enough to finish on the host (started), execute() / fold() calls performed
 * there, write-backs of the host's folded copies launched.  JOBS_FOLD: fold-only jobs of the group launches (arrays shared by
 * several claims or in none, folded inside the launch); CHAINS: groups of jobs run by one set of workgroups one after the other
 * (folds first, then the evaluations that read them back). */
enum { BN_GROUP_LAUNCHES = 0, BN_GROUP_JOBS_FUSED = 1, BN_GROUP_JOBS_EVAL = 2, BN_GROUP_PREFOLDS = 3, BN_GROUP_SPEC_JOBS = 4, BN_GROUP_SPEC_HITS = 5, BN_GROUP_EVALS = 6, BN_GROUP_FLUSHED_FOLDS = 7, BN_GROUP_HOSTED_STARTED = 8, BN_GROUP_HOSTED_EVALS = 9, BN_GROUP_HOSTED_FOLDS = 10, BN_GROUP_HOSTED_WRITEBACKS = 11, BN_GROUP_JOBS_FOLD = 12, BN_GROUP_CHAINS = 13, BN_GROUP_N = 14 };
int bn_group_counters(bn_ctx *ctx, uint64_t *counters /*[BN_GROUP_N]*/);

#ifdef __cplusplus
}
#endif
#endif /* BINIUS_AMD_H */
