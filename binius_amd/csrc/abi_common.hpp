// binius_amd/csrc/abi_common.hpp -- pieces shared by the translation units of the extern "C" boundary
// (abi.cpp: context, copies, deferral; abi_kernels.cpp: the recorded-kernel dispatcher; abi_ops.cpp: the
// executor ops, Merkle and NTT entry points).  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "internal.hpp"

using bn::f128;

// brackets the launches issued in its scope with two events when profiling is on
struct prof_scope {
	bn_ctx *ctx;
	int idx = -1;
	prof_scope(bn_ctx *c, int cls) : ctx(c)
	{
		if (!c->prof_on) return;
		auto get = [&]() {
			hipEvent_t e = nullptr;
			if (!c->ev_pool.empty()) {
				e = c->ev_pool.back();
				c->ev_pool.pop_back();
			} else {
				hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
			}
			return e;
		};
		bn_ctx::prof_rec r{cls, get(), get()};
		hipEventRecord(r.a, c->stream);
		c->prof.push_back(r);
		idx = (int)c->prof.size() - 1;
	}
	~prof_scope()
	{
		if (idx >= 0) hipEventRecord(ctx->prof[idx].b, ctx->stream);
	}
};

#define BN_REQUIRE(cond, msg)                                                  \
	do {                                                                       \
		if (!(cond))                                                           \
			return bn::fail(BN_ERR_INPUT_VALIDATION, std::string("input validation: ") + (msg)); \
	} while (0)


// BN_SHADOW_DEBUG=1: one line on stderr whenever the MLE-check shadow is made, used, missed or dropped (diagnostic)
inline bool shadow_debug()
{
	static const bool on = std::getenv("BN_SHADOW_DEBUG") != nullptr; // (a live diagnostic, not a measurement knob)
	return on;
}
#define BN_SHDBG(...)                         \
	do {                                      \
		if (shadow_debug()) {                 \
			fprintf(stderr, "[shadow] " __VA_ARGS__); \
			fputc('\n', stderr);              \
		}                                     \
	} while (0)

namespace bnabi {
// ---- deferral machinery (defined in abi.cpp)
int flush_copies(bn_ctx *ctx);
int flush_pending(bn_ctx *ctx, bool keep_tail = false, bool publish_tiny = false, bool keep_shadow = false);
int tail_cancel(bn_ctx *ctx);
std::vector<unsigned char> recipe_bytes(const bn::fin_args &a);
// resident tail kernel: command / status words in the pinned mailbox
inline volatile uint64_t *tail_cmd(bn_ctx *ctx) { return &ctx->h_mail[80].lo; }
inline volatile uint64_t *tail_status(bn_ctx *ctx) { return &ctx->h_mail[82].lo; }
// armed round: command block in the pinned mailbox (layout: arm.hpp)
inline volatile uint64_t *arm_cmd(bn_ctx *ctx) { return &ctx->h_mail[84].lo; }
inline volatile uint64_t *arm_status(bn_ctx *ctx) { return &ctx->h_mail[87].lo; }
void arm_cancel(bn_ctx *ctx);
// the context's side stream, ordered after everything enqueued on the main stream so far (first use since the last join)
hipStream_t side_stream(bn_ctx *ctx);
// launches what is queued for the side stream (bn_ctx::side_queue), in order
int side_run_queue(bn_ctx *ctx);
// the main stream waits for the side stream's work, queued work included (no host synchronisation)
void side_join(bn_ctx *ctx);
bool ranges_overlap(const void *p, uint64_t n_p, const void *q, uint64_t n_q);
bool independent_of_pending(bn_ctx *ctx, const void *p, uint64_t n);
// the MLE-check shadow's table: read eq[0], eq[2^k] to the host, derive the ratios rho_k, check the whole table against them
// (one gather + one pass + two stream synchronisations); false: not a tensor expansion (or a coordinate 0 / 1)
int shadow_check_table(bn_ctx *ctx, bool *ok);
// host tail (abi_kernels.cpp): launch the chain of folds the host performed on its own copy; the tail ends
int host_tail_flush(bn_ctx *ctx, bool publish);
// two deferred folds (two-round launches): run the first one now, the second becomes the deferred one
int flush_first_fold(bn_ctx *ctx);
// the deferred fold `pf` folds exactly the arrays the precomputed next-round sums describe
bool pre_matches(const bn_ctx::precomp_state &pre, const bn_ctx::pending_fold &pf);
// the product-sum kernels with the three-factor a * b * eq shape routed through two element-wise passes + the matrix-core
// kernel from 2^20 points (abi_kernels.cpp roundeval_product_routed); scratch_free: the context scratch is not in use
hipError_t roundeval_product_routed_pub(bn_ctx *ctx, bool scratch_free, const void *const *hi, const void *const *lo, uint32_t k, uint64_t n, bn::f128 *d_out);
// ---- arbitrary ArithCircuits on the throughput kernels (abi_circuit.cpp)
bool circuit_multipass_applies(const bn_ctx *ctx, const bn_expr *e, uint64_t row_len);
constexpr int kCircuitDeclined = -1000; // (internal: the planner declines this circuit -- run the interpreter)
// out[i] = expr(rows[.][i])
int circuit_multipass_map(bn_ctx *ctx, const bn_expr *e, const void *const *rows, uint64_t row_len, void *out, size_t scratch_off);
// temporaries of row_len elements circuit_multipass_sum will need (-1: declined)
int circuit_multipass_sum_temps(const bn_expr *e, bool has_eq);
// d_slots[0] ^ d_slots[1] ^= sum_i expr(rows[.][i]) * (eq ? eq[i] : 1)
int circuit_multipass_sum(bn_ctx *ctx, const bn_expr *e, const void *const *rows, uint64_t row_len, const void *eq, bn::f128 *d_slots, size_t scratch_off,
                          const void *ones_table = nullptr);
// ---- claim groups (abi_group.cpp)
// the single-claim deferral state only (pending fold(s), armed / resident kernels, host tail, shadow, deferred copies)
int flush_legacy(bn_ctx *ctx, bool keep_tail = false, bool publish_tiny = false, bool keep_shadow = false);
// is any of that state alive?
bool legacy_state_active(const bn_ctx *ctx);
// the pinned tables / accumulator slots / value mailbox of the group kernel's launches (allocated on first use)
int group_res_alloc(bn_ctx *ctx);
// a batch of folds can be deferred on the group path
bool group_fold_applies(const bn_ctx *ctx, uint32_t count, uint32_t scale_mask);
// [p, p + n) touches none of the arrays a deferred group fold reads or writes
bool group_independent(const bn_ctx *ctx, const void *p, uint64_t n, bool write = true);
// defer the batch (src0 | x1 -> x0, n elements each); deferred folds it overlaps run first
int group_defer_fold(bn_ctx *ctx, void *const *x0, const void *const *src0, const void *const *x1, uint32_t count, uint64_t n, f128 z);
// run every deferred group fold as a plain launch and forget every prediction
int group_flush(bn_ctx *ctx);
// run only the deferred folds that overlap [p, p + n); predictions that describe overlapping arrays are forgotten
int group_flush_touching(bn_ctx *ctx, const void *p, uint64_t n, bool publish_tiny = false, bool write = true);
// a host read of [p, p + n) inside one current array of a hosted prover (abi_group.cpp): answered from the host's copy (true)
bool group_host_read(bn_ctx *ctx, const void *p, uint64_t n, bn_f128 *h_dst);
// a write into [p, p + n) that flushed nothing: predictions that describe overlapping arrays are forgotten
void group_note_write(bn_ctx *ctx, const void *p, uint64_t n);
// the round evaluation of a prover (k product claims over m arrays) on the group path; *handled = false: not that shape / not
// wanted -- the caller goes on with the single-claim dispatcher
int group_eval(bn_ctx *ctx, const bn_memmap *maps, uint32_t n_maps, const bn_kop *ops, uint32_t n_ops, const uint32_t *ret_values, uint32_t n_ret,
               bn_f128 *h_out, bool *handled);
// the same plan over super-rows (n_b evaluation points side by side), its final sums collected as inner-product requests, and
// all requests of a call in one launch of the group kernel (abi_circuit.cpp)
struct ip_job {
	const void *a, *b; // b == null: the sum of row a
	bn::f128 coeff;
	uint32_t out;
};
struct ip_collector {
	std::vector<ip_job> jobs;
	std::vector<std::pair<uint32_t, bn::f128>> consts;
};
int circuit_multipass_collect(bn_ctx *ctx, const bn_expr *e, const void *const *rows, uint64_t seg, uint32_t n_b, const void *eq, size_t scratch_off,
                              const uint32_t *out_index, ip_collector &col);
int circuit_ip_run(bn_ctx *ctx, const ip_collector &col, uint64_t n, const void *ones, bn::f128 *values);
// ---- small helpers shared by the op entry points (abi.cpp)
int publish_result(bn_ctx *ctx, uint32_t n_groups, bn_f128 *h_out);
int publish_vals(bn_ctx *ctx, const bn::f128 *d_vals, uint32_t n_groups, uint32_t group_len, uint32_t g_stride, uint32_t i_stride, bn::f128 *h_out);
int upload_ptrs(bn_ctx *ctx, const void *const *ptrs, uint32_t n, const void ***d_ptrs);
int ensure_d_steps(const bn_expr *e);
} // namespace bnabi
using namespace bnabi;

// Every entry point also makes the context's device current on the calling thread: scratch buffers, NTT
// tables and pinned staging are allocated lazily inside calls, and a worker thread (or a process that
// drives several GPUs) would otherwise put them on whatever device that thread last used.
struct bn_enter_guard {
	std::lock_guard<std::recursive_mutex> lock;
	explicit bn_enter_guard(bn_ctx *c) : lock(c->mu) { (void)hipSetDevice(c->device); }
};
#define BN_ENTER(ctx) bn_enter_guard bn_enter_lock_(ctx)
#define BN_FLUSH(ctx)                    \
	do {                                 \
		int rc_ = flush_pending(ctx);    \
		(ctx)->mirror.valid = false;     \
		if (rc_) return rc_;             \
	} while (0)

// A call that touches device memory only inside the given ranges (and the context's own scratch): the single-claim state is
// flushed as by BN_FLUSH; of the claim groups' deferred folds only those that touch one of the ranges run, the others stay deferred
// for the next round's launch (FRIFolder::execute_fold_round between the folds and the next evaluations of a batch round,
// core/src/piop/prove.rs:372-384: fri_fold, the Merkle tree, its root).
struct bn_range {
	const void *p;
	uint64_t n; // field elements
};
static inline int flush_for_ranges(bn_ctx *ctx, std::initializer_list<bn_range> ranges)
{
	int rc = flush_legacy(ctx);
	ctx->mirror.valid = false;
	for (const bn_range &r : ranges)
		if (!rc && r.p && r.n) rc = group_flush_touching(ctx, r.p, r.n);
	return rc;
}
#define BN_FLUSH_FOR(ctx, ...)                                  \
	do {                                                        \
		int rc_ = flush_for_ranges(ctx, {__VA_ARGS__});         \
		if (rc_) return rc_;                                    \
	} while (0)

static inline bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }
static inline uint32_t ilog2(uint64_t n)
{
	uint32_t l = 0;
	while (n > 1) {
		n >>= 1;
		l++;
	}
	return l;
}
static inline f128 to_f(const bn_f128 *p) { return f128{p->lo, p->hi}; }
static inline bool valid_tower_level(uint32_t l) { return l == 0 || (l >= 3 && l <= 7); } // tower_macro.rs:9-15
