// binius_amd/csrc/abi_ops.cpp -- extern "C" entry points of the executor ops (tensor_expand, inner_product,
// fold_left / fold_right, fri_fold, compute_composite, pairwise_product_reduce), the Merkle / Groestl
// commitment side and the additive NTT: argument validation with the reference's error behaviour, then a
// kernel launch.  No arithmetic fallback lives here.
#include "abi_common.hpp"

extern "C" {

int bn_tensor_expand(bn_ctx *ctx, void *d_data, uint64_t data_len, uint32_t log_n, const bn_f128 *h_coords, uint32_t k)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(log_n + k < 64 && data_len == ((uint64_t)1 << (log_n + k)), "invalid data length");
	prof_scope ps(ctx, BN_PROF_TENSOR_EXPAND);
	std::vector<f128> coords(k);
	for (uint32_t i = 0; i < k; i++) coords[i] = to_f(&h_coords[i]);
	BN_HIP(bn::launch_tensor_expand(ctx->stream, ctx->n_cu, d_data, log_n, coords.data(), k));
	return BN_OK;
}

int bn_inner_product(bn_ctx *ctx, const void *d_a, uint64_t a_len, uint32_t tower_level, const void *d_b, uint64_t b_len,
                     bn_f128 *h_out)
{
	BN_REQUIRE(ctx && h_out, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(tower_level <= 7 && (a_len << (7 - tower_level)) == b_len, "invalid input: inner_product lengths");
	BN_REQUIRE(valid_tower_level(tower_level), "unsupported value of tower_level");
	ctx->s_clean = false; // slots 0..1 of the accumulator area are used as this op's accumulators
	BN_HIP(hipMemsetAsync(ctx->d_result, 0, 2 * sizeof(f128), ctx->stream));
	if (tower_level == 7 && b_len >= 2 && (b_len & 1) == 0) {
		// F x F: a plain sum of products -> the bit-sliced product-sum kernel (two half-range streams)
		if (bn::mfma_applies(ctx->n_cu, b_len / 2))
			BN_HIP(bn::launch_roundeval_mfma_split(ctx->stream, ctx->n_cu, d_a, d_b, b_len / 2, b_len / 2, ctx->d_result));
		else
			BN_HIP(bn::launch_roundeval9_split(ctx->stream, ctx->n_cu, d_a, d_b, b_len / 2, b_len / 2, ctx->d_result));
		return publish_result(ctx, 2, h_out);
	}
	if (tower_level == 5 && b_len >= 8192 && b_len % 512 == 0) {
		// B32 x F: four independent bit-sliced GF(2^32) inner products (kernels_ip32.hip)
		BN_HIP(bn::launch_ip32(ctx->stream, ctx->n_cu, d_a, d_b, b_len, ctx->d_result));
		return publish_result(ctx, 1, h_out);
	}
	BN_HIP(bn::launch_inner_product(ctx->stream, ctx->n_cu, d_a, tower_level, d_b, b_len, ctx->d_result));
	return publish_result(ctx, 1, h_out);
}

static int fold_common(bn_ctx *ctx, bool left, const void *d_mat, uint64_t mat_len, uint32_t tower_level, const void *d_vec,
                       uint64_t vec_len, void *d_out, uint64_t out_len)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(tower_level <= 7, "invalid evals: tower_level > 7");
	BN_REQUIRE(valid_tower_level(tower_level), "unsupported value of tower_level");
	BN_REQUIRE(is_pow2(mat_len) && is_pow2(vec_len), "lengths must be powers of two");
	const uint32_t log_evals = ilog2(mat_len) + 7 - tower_level;
	const uint32_t log_q = ilog2(vec_len);
	BN_REQUIRE(log_q <= log_evals, "query larger than evals");
	BN_REQUIRE(out_len == ((uint64_t)1 << (log_evals - log_q)), "output has the wrong number of elements");
	if (left) {
		if (tower_level == 5 && out_len >= 4096 && (vec_len == 16 || vec_len == 32 || vec_len == 64)) {
			void *tab = bn::ctx_scratch(ctx, bn::linmap_table_bytes(vec_len * 32));
			if (tab) {
				const hipError_t e = bn::launch_fold_left_mfma(ctx->stream, ctx->n_cu, d_mat, tower_level, d_vec, vec_len, d_out, out_len, tab);
				if (e == hipSuccess) return BN_OK;
				if (e != hipErrorNotSupported) BN_HIP(e);
			}
		}
		BN_HIP(bn::launch_fold_left(ctx->stream, ctx->n_cu, d_mat, tower_level, d_vec, vec_len, d_out, out_len));
		return BN_OK;
	}
	// rows of 512 .. 2048 bits: the linear map on the matrix cores (kernels_linmap.hip); every other shape: nibble tables
	const uint64_t row_bits = vec_len << tower_level;
	if (out_len >= 4096 && (row_bits == 512 || row_bits == 1024 || row_bits == 2048)) {
		void *tab = bn::ctx_scratch(ctx, bn::linmap_table_bytes(row_bits));
		if (tab) {
			const hipError_t e = bn::launch_fold_right_mfma(ctx->stream, ctx->n_cu, d_mat, tower_level, d_vec, vec_len, d_out, out_len, tab);
			if (e == hipSuccess) return BN_OK;
			if (e != hipErrorNotSupported) BN_HIP(e);
		}
	}
	BN_HIP(bn::launch_fold_right(ctx->stream, ctx->n_cu, d_mat, tower_level, d_vec, vec_len, d_out, out_len));
	return BN_OK;
}

int bn_fold_left(bn_ctx *ctx, const void *d_mat, uint64_t mat_len, uint32_t tower_level, const void *d_vec, uint64_t vec_len,
                 void *d_out, uint64_t out_len)
{
	return fold_common(ctx, true, d_mat, mat_len, tower_level, d_vec, vec_len, d_out, out_len);
}

int bn_fold_right(bn_ctx *ctx, const void *d_mat, uint64_t mat_len, uint32_t tower_level, const void *d_vec, uint64_t vec_len,
                  void *d_out, uint64_t out_len)
{
	return fold_common(ctx, false, d_mat, mat_len, tower_level, d_vec, vec_len, d_out, out_len);
}

// The twiddle basis of the caller's NTT instance on the device (d_out) + `extra_bytes` of scratch (extra).  The basis is
// caller-owned pageable memory that does not change between the calls of one instance: it is compared with the copy of the
// last call (8 KiB) and uploaded -- one synchronisation -- only when it differs.
static int upload_s_evals(bn_ctx *ctx, const uint64_t *h_s_evals, uint64_t **d_out, size_t extra_bytes, void **extra)
{
	constexpr size_t words = (size_t)BN_NTT_MAX_DIM * BN_NTT_MAX_DIM;
	if (ctx->h_s_evals.size() != words || memcmp(ctx->h_s_evals.data(), h_s_evals, words * sizeof(uint64_t)) != 0) {
		// (kernels of earlier calls that still read the old basis are ahead of the copy on the same stream)
		// (the host copy only names what IS on the device: forgotten first, so that a failed upload is retried by the next call)
		ctx->h_s_evals.clear();
		BN_HIP(hipMemcpyAsync(ctx->d_s_evals, h_s_evals, words * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
		BN_HIP(hipStreamSynchronize(ctx->stream)); // (the source is pageable: the runtime may still be reading it)
		ctx->h_s_evals.assign(h_s_evals, h_s_evals + words);
	}
	*d_out = ctx->d_s_evals;
	if (extra) {
		char *scr = (char *)bn::ctx_scratch(ctx, extra_bytes);
		if (!scr)
			return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (scratch)");
		*extra = scr;
	}
	return BN_OK;
}

int bn_fri_fold(bn_ctx *ctx, const uint64_t *h_s_evals, uint32_t tw_level, uint32_t log_domain, uint32_t log_len,
                uint32_t log_batch_size, const bn_f128 *h_challenges, uint32_t n_challenges, const void *d_in, uint64_t in_len,
                void *d_out, uint64_t out_len)
{
	BN_REQUIRE(ctx && h_s_evals, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH_FOR(ctx, bn_range{d_in, in_len}, bn_range{d_out, out_len}); // (deferred folds of a batch's provers that touch neither stay deferred)
	BN_REQUIRE(log_len + log_batch_size < 64 && in_len == ((uint64_t)1 << (log_len + log_batch_size)), "invalid data_in length");
	BN_REQUIRE(n_challenges >= log_batch_size, "invalid challenges length");
	BN_REQUIRE(n_challenges <= log_batch_size + log_len, "challenges length too big");
	BN_REQUIRE(out_len == ((uint64_t)1 << (log_len - (n_challenges - log_batch_size))), "invalid data_out length");
	BN_REQUIRE(tw_level >= 3 && tw_level <= 6, "unsupported twiddle field");
	BN_REQUIRE(log_len <= log_domain && log_domain <= BN_NTT_MAX_DIM, "NTT domain too small");
	if (n_challenges == 0) {
		BN_HIP(hipMemcpyAsync(d_out, d_in, in_len * sizeof(f128), hipMemcpyDeviceToDevice, ctx->stream));
		return BN_OK;
	}
	uint64_t *d_s = nullptr;
	void *pp = nullptr;
	int rc = upload_s_evals(ctx, h_s_evals, &d_s, in_len * sizeof(f128), &pp);
	if (rc) return rc;
	std::vector<f128> ch(n_challenges);
	for (uint32_t i = 0; i < n_challenges; i++) ch[i] = to_f(&h_challenges[i]);
	BN_HIP(bn::launch_fri_fold(ctx->stream, d_s, tw_level, log_domain, log_len, log_batch_size, ch.data(), n_challenges, d_in,
	                           d_out, out_len, pp, ctx->n_cu, ctx->d_mul8));
	return BN_OK;
}

// rows -> device array of row pointers, staged in the tail of the result mailbox area
int bn_compute_composite(bn_ctx *ctx, const void *const *d_rows, uint32_t n_rows, uint64_t row_len, void *d_out,
                         uint64_t out_len, const bn_expr *expr)
{
	BN_REQUIRE(ctx && expr, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(row_len == out_len, "inputs and output must be the same length");
	BN_REQUIRE(expr->n_vars == n_rows || (expr->n_vars <= n_rows), "composition not match with input");
	BN_REQUIRE(expr->steps.size() <= 64, "circuit too large for this backend (max 64 steps)");
	if (expr->shape == bn_expr::PRODUCT && expr->product_vars.size() == 2) {
		BN_HIP(bn::launch_mul9(ctx->stream, ctx->n_cu, d_rows[expr->product_vars[0]], 1, d_rows[expr->product_vars[1]], 1, 0, d_out,
		                       row_len));
		return BN_OK;
	}
	int rc;
	if (circuit_multipass_applies(ctx, expr, row_len)) {
		// any other circuit: compiled into passes of the throughput kernels (abi_circuit.cpp)
		rc = circuit_multipass_map(ctx, expr, d_rows, row_len, d_out, 0);
		if (rc != kCircuitDeclined) return rc;
	}
	const void **d_ptrs = nullptr;
	rc = upload_ptrs(ctx, d_rows, n_rows, &d_ptrs);
	if (rc) return rc;
	rc = ensure_d_steps(expr);
	if (rc) return rc;
	BN_HIP(bn::launch_compute_composite_generic(ctx->stream, d_ptrs, n_rows, row_len, d_out, expr->d_steps,
	                                            (uint32_t)expr->steps.size()));
	return BN_OK;
}

int bn_pairwise_product_reduce(bn_ctx *ctx, const void *d_in, uint64_t n, void *const *d_round_outs, const uint64_t *round_lens,
                               uint32_t n_rounds)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(is_pow2(n), "input length must be a power of 2");
	BN_REQUIRE(n >= 2, "input length must be greater than or equal to 2 in order to perform at least one reduction");
	const uint32_t log_n = ilog2(n);
	BN_REQUIRE(n_rounds == log_n, "round_outputs.len() does not match the expected length");
	for (uint32_t r = 0; r < n_rounds; r++)
		BN_REQUIRE(round_lens[r] == ((uint64_t)1 << (log_n - r - 1)), "round_outputs[i].len() has the wrong size");
	// Large levels: the element-wise product (kernels_mul9.hip, at throughput), up to BN_MUL9_FUSE (2) levels per launch
	// (k_mul9_tree: a workgroup goes on with the products under the 896 it has just stored).  From 2^15 elements down a
	// level is one dependent chain whatever its size, and the tree is walked by workgroups that keep their subtree on the CU
	// (kernels_pairtree.hip): up to BN_PAIRTREE_LOG_S (6) levels per launch.  (Round 2 measured the bit-sliced product itself inside one
	// workgroup for the last 12 levels: 233 vs 225 us -- its chain is the cost)
	static const uint32_t tree_max_log2 = [] {
		const char *e = bn::settled_knob("BN_PAIRTREE_MAX_LOG2"); // measurement knob: 0 = off
		const int v = e ? atoi(e) : 15;
		return (uint32_t)(v < 0 ? 0 : (v > 24 ? 24 : v));
	}();
	static const uint32_t tree_log_s = [] {
		const char *e = bn::settled_knob("BN_PAIRTREE_LOG_S");
		const int v = e ? atoi(e) : 6;
		return (uint32_t)(v < 1 ? 1 : (v > 8 ? 8 : v));
	}();
	const void *src = d_in;
	uint32_t r = 0;
	static const uint32_t fuse = [] {
		const char *e = bn::settled_knob("BN_MUL9_FUSE"); // levels per launch of the element-wise kernel (1 .. 4)
		const int v = e ? atoi(e) : 2;
		return (uint32_t)(v < 1 ? 1 : (v > 4 ? 4 : v));
	}();
	while (r < n_rounds && log_n - r > tree_max_log2) {
		uint32_t k = log_n - r - tree_max_log2; // levels left for this kernel
		if (k > fuse) k = fuse;
		if (k == 1) {
			BN_HIP(bn::launch_mul9(ctx->stream, ctx->n_cu, src, 2, src, 2, 1, d_round_outs[r], round_lens[r]));
		} else {
			bn::mul9_tree_args ta{};
			ta.in = (const uint32_t *)src;
			ta.n0 = round_lens[r];
			ta.n_levels = k;
			for (uint32_t l = 0; l < k; l++) ta.lv[l] = (uint32_t *)d_round_outs[r + l];
			BN_HIP(bn::launch_mul9_tree(ctx->stream, ctx->n_cu, ta));
		}
		r += k;
		src = d_round_outs[r - 1];
	}
	while (r < n_rounds) {
		const uint32_t log_m = log_n - r; // elements of `src`
		const uint32_t nl = log_m < tree_log_s ? log_m : tree_log_s;
		bn::pairtree_args pa{};
		pa.in = (const f128 *)src;
		pa.n_levels = nl;
		for (uint32_t l = 0; l < nl; l++) pa.out[l] = (f128 *)d_round_outs[r + l];
		BN_HIP(bn::launch_pairtree(ctx->stream, pa, tree_log_s, (uint64_t)1 << (log_m - nl)));
		r += nl;
		src = d_round_outs[r - 1];
	}
	return BN_OK;
}

// ---------------------------------------------------------------------------------- Merkle / Groestl
int bn_groestl256_leaves(bn_ctx *ctx, const void *d_elems, uint64_t n_elems, uint64_t batch_size, void *d_digests)
{
	BN_REQUIRE(ctx && d_elems && d_digests, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(batch_size != 0 && n_elems % batch_size == 0, "IncorrectBatchSize");
	BN_HIP(bn::launch_groestl_leaves(ctx->stream, ctx->n_cu, d_elems, batch_size, n_elems / batch_size, d_digests));
	return BN_OK;
}

int bn_groestl256_compress_layer(bn_ctx *ctx, const void *d_prev, uint64_t n_out, void *d_next)
{
	BN_REQUIRE(ctx && d_prev && d_next, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_HIP(bn::launch_groestl_layer(ctx->stream, ctx->n_cu, d_prev, n_out, d_next));
	return BN_OK;
}

int bn_merkle_build(bn_ctx *ctx, const void *d_elems, uint64_t n_elems, uint64_t batch_size, void *d_nodes)
{
	BN_REQUIRE(ctx && d_elems && d_nodes, "null argument");
	BN_ENTER(ctx);
	BN_REQUIRE(batch_size != 0 && n_elems % batch_size == 0, "IncorrectBatchSize");
	BN_FLUSH_FOR(ctx, bn_range{d_elems, n_elems}, bn_range{d_nodes, 2 * (2 * (n_elems / batch_size) - 1)});
	const uint64_t n_leaves = n_elems / batch_size;
	BN_REQUIRE(n_leaves != 0 && (n_leaves & (n_leaves - 1)) == 0, "PowerOfTwoLengthRequired");
	BN_HIP(bn::launch_groestl_leaves(ctx->stream, ctx->n_cu, d_elems, batch_size, n_leaves, d_nodes));
	BN_HIP(bn::launch_merkle_layers(ctx->stream, ctx->n_cu, d_nodes, n_leaves));
	return BN_OK;
}

// Openings: h_out[i * item_elems .. +item_elems) = d_src[h_offsets[i] .. +item_elems).  One kernel reads the
// offsets from and writes the items to pinned host memory; one synchronisation.
int bn_gather_d2h(bn_ctx *ctx, const void *d_src, const uint64_t *h_offsets, uint64_t n_items, uint64_t item_elems, bn_f128 *h_out)
{
	BN_REQUIRE(ctx && d_src && (n_items == 0 || (h_offsets && h_out)), "null argument");
	BN_ENTER(ctx);
	{
		uint64_t hi = 0; // (the items lie inside [d_src, d_src + max offset + item_elems))
		for (uint64_t i = 0; i < n_items; i++)
			if (h_offsets[i] > hi) hi = h_offsets[i];
		BN_FLUSH_FOR(ctx, bn_range{d_src, hi + item_elems});
	}
	if (n_items == 0 || item_elems == 0) return BN_OK;
	BN_REQUIRE(n_items <= (1ull << 24) && item_elems <= (1ull << 24) && n_items * item_elems <= (1ull << 26), "gather: too many elements for one call");
	if (n_items == 1 && item_elems <= 64) {
		// one short item -- the root of a tree just built (MerkleTreeProver::commit reads it every FRI commit round,
		// fri/prove.rs:395-420): through the zero-copy mailbox, a spin on the sequence word instead of a stream synchronisation
		// (flush_for_ranges above left no mirrored fold in the mailbox)
		f128 vals[64];
		const int rc = publish_vals(ctx, (const f128 *)d_src + h_offsets[0], 1, (uint32_t)item_elems, 0, 1, vals);
		if (rc) return rc;
		for (uint64_t e = 0; e < item_elems; e++) h_out[e] = bn_f128{vals[e].lo, vals[e].hi};
		return BN_OK;
	}
	const size_t off_bytes = ((size_t)n_items * 8 + 15) & ~(size_t)15;
	const size_t need = off_bytes + (size_t)n_items * item_elems * sizeof(f128);
	if (need > ctx->gather_bytes) {
		BN_HIP(hipStreamSynchronize(ctx->stream));
		if (ctx->h_gather) hipHostFree(ctx->h_gather);
		ctx->h_gather = nullptr;
		ctx->gather_bytes = 0;
		size_t cap = 1 << 16;
		while (cap < need) cap <<= 1;
		if (hipHostMalloc(&ctx->h_gather, cap, hipHostMallocMapped) != hipSuccess)
			return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (pinned gather buffer)");
		BN_HIP(hipHostGetDevicePointer(&ctx->d_gather, ctx->h_gather, 0));
		ctx->gather_bytes = cap;
	}
	std::memcpy(ctx->h_gather, h_offsets, (size_t)n_items * 8);
	BN_HIP(bn::launch_gather(ctx->stream, d_src, (const uint64_t *)ctx->d_gather, n_items, item_elems, (char *)ctx->d_gather + off_bytes));
	BN_HIP(hipStreamSynchronize(ctx->stream));
	std::memcpy(h_out, (const char *)ctx->h_gather + off_bytes, (size_t)n_items * item_elems * sizeof(f128));
	return BN_OK;
}

// ---------------------------------------------------------------------------------- NTT
static int ntt_common(bn_ctx *ctx, bool inverse, void *d_data, uint32_t elem_level, uint32_t tw_level,
                      const uint64_t *h_s_evals, uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z,
                      uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds)
{
	BN_REQUIRE(ctx && h_s_evals, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(elem_level >= 3 && elem_level <= 7, "unsupported element field");
	BN_REQUIRE(tw_level >= 3 && tw_level <= 6 && tw_level <= elem_level, "unsupported twiddle field");
	BN_REQUIRE(log_domain >= 1 && log_domain <= BN_NTT_MAX_DIM && log_domain <= (1u << tw_level), "bad NTT domain");
	BN_REQUIRE(coset_bits >= 64 || coset < ((uint64_t)1 << coset_bits), "coset index out of bounds");
	BN_REQUIRE(log_y + coset_bits <= log_domain, "NTT domain too small");
	BN_REQUIRE(skip_rounds <= log_y, "skip_rounds larger than log_y");
	BN_REQUIRE(log_x + log_y + log_z < 48, "transform too large");
	if (log_y == 0 || skip_rounds == log_y) return BN_OK;
	if (elem_level >= 5 && tw_level == 5 && log_y >= 14 && !bn::settled_knob("BN_NTT_NO_BITSLICE")) {
		// large transforms with B32 twiddles: bit-sliced butterflies (kernels_ntt_bs.hip); B64 / B128
		// data and log_x / log_z batches are interleaved B32 transforms
		const uint32_t lx = log_x + (elem_level - 5);
		void *scr = bn::ctx_scratch(ctx, bn::ntt_bs_scratch_bytes(log_y + lx + log_z));
		if (!scr) return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (NTT scratch)");
		if (!ctx->ntt_cache) {
			bn::ntt_bs_cache *nc = new bn::ntt_bs_cache;
			if (hipMalloc(&nc->d_tables, bn::ntt_bs_tables_bytes()) != hipSuccess) {
				delete nc;
				return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (NTT tables)");
			}
			ctx->ntt_cache = nc;
		}
		prof_scope ps(ctx, BN_PROF_NTT);
		hipError_t be = bn::launch_ntt_bs(ctx->stream, inverse, d_data, h_s_evals, log_domain, lx, log_y, log_z, coset, coset_bits,
		                                  skip_rounds, scr, (bn::ntt_bs_cache *)ctx->ntt_cache);
		if (be == hipSuccess) return BN_OK;
		if (be != hipErrorNotSupported) return bn::hip_fail(be, "launch_ntt_bs");
	}
	uint64_t *d_s = nullptr;
	int rc = upload_s_evals(ctx, h_s_evals, &d_s, 0, nullptr);
	if (rc) return rc;
	prof_scope ps(ctx, BN_PROF_NTT);
	if (!bn::settled_knob("BN_NTT_PER_LAYER")) {
		hipError_t te = bn::launch_ntt_tiled(ctx->stream, ctx->n_cu, inverse, d_data, elem_level, tw_level, ctx->d_mul8, d_s, log_domain,
		                                     log_x, log_y, log_z, coset, coset_bits, skip_rounds);
		if (te == hipSuccess) return BN_OK;
		if (te != hipErrorNotSupported) return bn::hip_fail(te, "launch_ntt_tiled");
	}
	BN_HIP(bn::launch_ntt(ctx->stream, inverse, d_data, elem_level, tw_level, d_s, log_domain, log_x, log_y, log_z, coset,
	                      coset_bits, skip_rounds));
	return BN_OK;
}

int bn_ntt_forward(bn_ctx *ctx, void *d_data, uint32_t elem_level, uint32_t tw_level, const uint64_t *h_s_evals,
                   uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset, uint32_t coset_bits,
                   uint32_t skip_rounds)
{
	return ntt_common(ctx, false, d_data, elem_level, tw_level, h_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits,
	                  skip_rounds);
}

int bn_ntt_inverse(bn_ctx *ctx, void *d_data, uint32_t elem_level, uint32_t tw_level, const uint64_t *h_s_evals,
                   uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z, uint64_t coset, uint32_t coset_bits,
                   uint32_t skip_rounds)
{
	return ntt_common(ctx, true, d_data, elem_level, tw_level, h_s_evals, log_domain, log_x, log_y, log_z, coset, coset_bits,
	                  skip_rounds);
}

// OnTheFlyTwiddleAccess::generate over BinarySubspace::with_dim(log_domain)
// (crates/ntt/src/twiddle.rs:107-124, 244-306; crates/math/src/binary_subspace.rs:33-38).
// O(log_domain^2) field operations of host metadata; uses the same gf128.hpp arithmetic as the
// kernels (subfield elements embed into the low bits).
int bn_ntt_s_evals(uint32_t tw_level, uint32_t log_domain, uint64_t *h_s_evals)
{
	BN_REQUIRE(h_s_evals, "null argument");
	BN_REQUIRE(tw_level >= 3 && tw_level <= 6, "unsupported twiddle field");
	BN_REQUIRE(log_domain >= 1 && log_domain <= BN_NTT_MAX_DIM && log_domain <= (1u << tw_level), "bad NTT domain");
	auto mul = [](uint64_t a, uint64_t b) { return bn::mul_slow(f128{a, 0}, f128{b, 0}).lo; };
	const uint32_t bits = 1u << tw_level;
	auto inv = [&](uint64_t a) {
		// a^(2^bits - 2) by square-and-multiply: prod_{i=1}^{bits-1} a^(2^i)
		uint64_t r = 1, sq = a;
		for (uint32_t i = 1; i < bits; i++) {
			sq = mul(sq, sq);
			r = mul(r, sq);
		}
		return r;
	};
	auto subspace_map = [&](uint64_t e, uint64_t c) { return mul(e, e) ^ mul(c, e); };
	const uint32_t d = log_domain;
	std::memset(h_s_evals, 0, sizeof(uint64_t) * BN_NTT_MAX_DIM * BN_NTT_MAX_DIM);
	std::vector<uint64_t> norm(d);
	norm[0] = 1;
	for (uint32_t b = 0; b + 1 < d; b++) h_s_evals[b] = 1ull << (b + 1);
	for (uint32_t i = 1; i < d; i++) {
		const uint64_t *prev = &h_s_evals[(i - 1) * BN_NTT_MAX_DIM];
		uint64_t *cur = &h_s_evals[i * BN_NTT_MAX_DIM];
		norm[i] = subspace_map(prev[0], norm[i - 1]);
		for (uint32_t b = 0; b + 1 + i < d; b++) cur[b] = subspace_map(prev[b + 1], norm[i - 1]);
	}
	for (uint32_t i = 0; i < d; i++) {
		const uint64_t iv = inv(norm[i]);
		uint64_t *cur = &h_s_evals[i * BN_NTT_MAX_DIM];
		for (uint32_t b = 0; b + 1 + i < d; b++) cur[b] = mul(cur[b], iv);
	}
	return BN_OK;
}

} // extern "C"
