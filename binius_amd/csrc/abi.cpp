// binius_amd/csrc/abi.cpp -- the extern "C" boundary declared in include/binius_amd.h:
// argument validation with the reference's error behaviour, context / stream / scratch
// management, the recorded-kernel dispatcher behind accumulate_kernels / map_kernels, and the
// host-only helpers (log_chunks_range, twiddle basis generation).
//
// No arithmetic fallback lives here: every op is a HIP kernel launch.  Host-side field arithmetic
// is used only for O(log n) metadata (twiddle basis = OnTheFlyTwiddleAccess::generate).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "internal.hpp"

using bn::f128;

// ---------------------------------------------------------------------------------- errors
namespace {
thread_local std::string g_last_error;
}

namespace bn {
void set_error(const std::string &msg) { g_last_error = msg; }
int fail(int code, const std::string &msg)
{
	g_last_error = msg;
	return code;
}
int hip_fail(hipError_t e, const char *what)
{
	std::ostringstream os;
	os << "device error: " << hipGetErrorString(e) << " (" << what << ")";
	g_last_error = os.str();
	return BN_ERR_DEVICE;
}
void *ctx_scratch(bn_ctx *ctx, size_t bytes)
{
	if (bytes <= ctx->scratch_bytes)
		return ctx->scratch;
	// the previous buffer may still be in use by enqueued work
	hipStreamSynchronize(ctx->stream);
	if (ctx->scratch)
		hipFree(ctx->scratch);
	ctx->scratch = nullptr;
	ctx->scratch_bytes = 0;
	size_t want = bytes + (bytes >> 2);
	if (hipMalloc(&ctx->scratch, want) != hipSuccess) {
		if (hipMalloc(&ctx->scratch, bytes) != hipSuccess)
			return nullptr;
		want = bytes;
	}
	ctx->scratch_bytes = want;
	return ctx->scratch;
}
} // namespace bn

namespace {
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
} // namespace

#define BN_REQUIRE(cond, msg)                                                  \
	do {                                                                       \
		if (!(cond))                                                           \
			return bn::fail(BN_ERR_INPUT_VALIDATION, std::string("input validation: ") + (msg)); \
	} while (0)

// Launch a deferred extrapolate_line batch (see bn_extrapolate_line_batch).  Every entry point
// that can observe device memory or the stream calls this first, so the deferral is invisible.
// ---- resident tail kernel: host side of the protocol (device side: kernels_foldeval9.hip)
// command block in the pinned mailbox page: h_mail[80].lo = command word, h_mail[81] = z,
// h_mail[82].lo = status (the id of the last tail kernel that exited)
static volatile uint64_t *tail_cmd(bn_ctx *ctx) { return &ctx->h_mail[80].lo; }
static volatile uint64_t *tail_status(bn_ctx *ctx) { return &ctx->h_mail[82].lo; }

// Stop the resident kernel (if any) and wait until it has left the device.
static int tail_cancel(bn_ctx *ctx)
{
	if (!ctx->tail.active) return BN_OK;
	ctx->tail.active = false;
	__atomic_store_n(tail_cmd(ctx), (ctx->tail.id << 20) | 0xFFFFFull, __ATOMIC_RELEASE);
	BN_HIP(hipStreamSynchronize(ctx->stream));
	return BN_OK;
}

static std::vector<unsigned char> recipe_bytes(const bn::fin_args &a)
{
	bn::fin_args r;
	std::memset(&r, 0, sizeof(r));
	r.n_terms = a.n_terms;
	r.n_values = a.n_values;
	r.n_ret = a.n_ret;
	r.n_slots = a.n_slots;
	for (uint32_t t = 0; t < a.n_terms; t++) {
		r.terms[t].value = a.terms[t].value;
		r.terms[t].slot = a.terms[t].slot;
		r.terms[t].coeff = a.terms[t].coeff;
	}
	for (uint32_t v = 0; v < a.n_values; v++) r.init[v] = a.init[v];
	for (uint32_t i = 0; i < a.n_ret; i++) r.ret_ids[i] = a.ret_ids[i];
	const unsigned char *p = reinterpret_cast<const unsigned char *>(&r);
	return std::vector<unsigned char>(p, p + sizeof(r));
}

static int flush_copies(bn_ctx *ctx)
{
	for (const auto &c : ctx->pend_copies)
		BN_HIP(hipMemcpyAsync(c.dst, c.src, c.n * sizeof(f128), hipMemcpyDeviceToDevice, ctx->stream));
	ctx->pend_copies.clear();
	return BN_OK;
}

static int flush_pending(bn_ctx *ctx, bool keep_tail = false, bool publish_tiny = false)
{
	if (ctx->tail.active && !keep_tail) {
		int rc = tail_cancel(ctx);
		if (rc) return rc;
	}
	if (!ctx->pend_copies.empty()) {
		int rc = flush_copies(ctx);
		if (rc) return rc;
	}
	if (!ctx->pend.active) return BN_OK;
	ctx->pend.active = false;
	if (publish_tiny && (uint64_t)ctx->pend.count * ctx->pend.n <= 64) {
		// the caller is a host read: fold and mirror the (few) results into the mailbox in one launch
		const uint64_t seq = ++ctx->mail_seq;
		prof_scope ps(ctx, BN_PROF_FOLD);
		BN_HIP(bn::launch_fold_publish(ctx->stream, ctx->pend.x0, ctx->pend.src0, ctx->pend.x1, ctx->pend.count, (uint32_t)ctx->pend.n,
		                               ctx->pend.z, ctx->d_mail, seq));
		ctx->mirror.valid = true;
		ctx->mirror.seq = seq;
		ctx->mirror.count = ctx->pend.count;
		ctx->mirror.n = (uint32_t)ctx->pend.n;
		for (uint32_t i = 0; i < ctx->pend.count; i++) ctx->mirror.ptr[i] = ctx->pend.x0[i];
		return BN_OK;
	}
	bn::fold_batch fb{};
	for (uint32_t i = 0; i < ctx->pend.count; i++) {
		fb.x0[i] = ctx->pend.x0[i];
		fb.x1[i] = ctx->pend.x1[i];
		if (ctx->pend.src0[i] != ctx->pend.x0[i]) // absorbed copy: materialise it, then fold in place
			BN_HIP(hipMemcpyAsync(ctx->pend.x0[i], ctx->pend.src0[i], ctx->pend.n * sizeof(f128), hipMemcpyDeviceToDevice, ctx->stream));
	}
	prof_scope ps(ctx, BN_PROF_FOLD);
	BN_HIP(bn::launch_extrapolate_line_batch(ctx->stream, ctx->n_cu, fb, ctx->pend.count, ctx->pend.n, ctx->pend.z));
	return BN_OK;
}
// one call at a time per context (the trait allows the host to call from several threads: rayon join/map)
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

static bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }
static uint32_t ilog2(uint64_t n)
{
	uint32_t l = 0;
	while (n > 1) {
		n >>= 1;
		l++;
	}
	return l;
}
static f128 to_f(const bn_f128 *p) { return f128{p->lo, p->hi}; }
static bool valid_tower_level(uint32_t l) { return l == 0 || (l >= 3 && l <= 7); } // tower_macro.rs:9-15

extern "C" {

const char *bn_last_error(void) { return g_last_error.c_str(); }
const char *bn_version(void) { return "binius_amd gfx950 abi-1"; }

// ---------------------------------------------------------------------------------- context
int bn_ctx_create(int device, uint64_t arena_elems, bn_ctx **out)
{
	if (!out)
		return bn::fail(BN_ERR_INPUT_VALIDATION, "input validation: out is null");
	int n_dev = 0;
	BN_HIP(hipGetDeviceCount(&n_dev));
	if (n_dev == 0)
		return bn::fail(BN_ERR_DEVICE, "device error: no HIP device visible (this backend has no CPU fallback)");
	BN_REQUIRE(device >= 0 && device < n_dev, "device ordinal out of range");
	BN_HIP(hipSetDevice(device));
	hipDeviceProp_t prop;
	BN_HIP(hipGetDeviceProperties(&prop, device));
	bn_ctx *ctx = new bn_ctx();
	ctx->device = device;
	ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	BN_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
	ctx->own_stream = true;
	BN_HIP(hipEventCreate(&ctx->ev0));
	BN_HIP(hipEventCreate(&ctx->ev1));
	BN_HIP(hipMalloc((void **)&ctx->d_result, sizeof(f128) * bn::kResultSlots));
	BN_HIP(hipHostMalloc((void **)&ctx->h_result, sizeof(f128) * bn::kResultSlots, hipHostMallocDefault));
	BN_HIP(hipHostMalloc((void **)&ctx->h_mail, sizeof(f128) * 128, hipHostMallocMapped | hipHostMallocCoherent));
	std::memset(ctx->h_mail, 0, sizeof(f128) * 128);
	BN_HIP(hipHostGetDevicePointer((void **)&ctx->d_mail, ctx->h_mail, 0));
	BN_HIP(hipMemset(ctx->d_result, 0, sizeof(f128) * bn::kResultSlots));
	ctx->s_clean = true;
	BN_HIP(hipMalloc((void **)&ctx->d_ticket, sizeof(unsigned)));
	BN_HIP(hipMemset(ctx->d_ticket, 0, sizeof(unsigned)));
	BN_HIP(hipMalloc((void **)&ctx->d_mul8, 65536));
	BN_HIP(bn::launch_build_mul8(ctx->stream, ctx->d_mul8));
	if (arena_elems) {
		hipError_t e = hipMalloc(&ctx->arena, arena_elems * sizeof(f128));
		if (e != hipSuccess) {
			bn_ctx_destroy(ctx);
			return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (device arena)");
		}
		ctx->arena_elems = arena_elems;
	}
	ctx->lazy_fold = getenv("BN_NO_LAZY_FOLD") == nullptr;
	if (const char *t = getenv("BN_TAIL_MAX_LOG2")) {
		const int l = atoi(t);
		ctx->tail_max_n_in = (l >= 3 && l <= 12) ? (1ull << l) : 0; // one workgroup: at most 2^12 elements per array
	}
	if (!ctx->lazy_fold) ctx->tail_max_n_in = 0;
	*out = ctx;
	return BN_OK;
}

int bn_ctx_destroy(bn_ctx *ctx)
{
	if (!ctx)
		return BN_OK;
	hipSetDevice(ctx->device);
	if (ctx->stream)
		hipStreamSynchronize(ctx->stream);
	if (ctx->ntt_cache) {
		bn::ntt_bs_cache *nc = (bn::ntt_bs_cache *)ctx->ntt_cache;
		if (nc->d_tables) hipFree(nc->d_tables);
		delete nc;
	}
	if (ctx->arena) hipFree(ctx->arena);
	if (ctx->scratch) hipFree(ctx->scratch);
	if (ctx->d_result) hipFree(ctx->d_result);
	if (ctx->d_mul8) hipFree(ctx->d_mul8);
	if (ctx->d_ticket) hipFree(ctx->d_ticket);
	if (ctx->h_result) hipHostFree(ctx->h_result);
	if (ctx->h_mail) hipHostFree(ctx->h_mail);
	if (ctx->h_gather) hipHostFree(ctx->h_gather);
	if (ctx->ev0) hipEventDestroy(ctx->ev0);
	if (ctx->ev1) hipEventDestroy(ctx->ev1);
	for (auto &r : ctx->prof) {
		hipEventDestroy(r.a);
		hipEventDestroy(r.b);
	}
	for (auto e : ctx->ev_pool) hipEventDestroy(e);
	if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
	delete ctx;
	return BN_OK;
}

int bn_arena_base(bn_ctx *ctx, void **d_base, uint64_t *elems)
{
	BN_REQUIRE(ctx && d_base && elems, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	*d_base = ctx->arena;
	*elems = ctx->arena_elems;
	return BN_OK;
}

int bn_ctx_set_stream(bn_ctx *ctx, void *hip_stream)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_HIP(hipSetDevice(ctx->device));
	BN_HIP(hipStreamSynchronize(ctx->stream));
	if (hip_stream == nullptr) {
		if (!ctx->own_stream) {
			BN_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
			ctx->own_stream = true;
			ctx->lazy_fold = getenv("BN_NO_LAZY_FOLD") == nullptr;
		}
		return BN_OK;
	}
	if (ctx->own_stream)
		hipStreamDestroy(ctx->stream);
	ctx->stream = (hipStream_t)hip_stream;
	ctx->own_stream = false;
	// On a stream the caller also enqueues on, deferred folds/copies would be visible: work the caller
	// puts on the stream right after bn_extrapolate_line would run BEFORE the fold.  Deferral is
	// therefore off on caller-supplied streams unless the caller opts in (BN_LAZY_ON_SHARED_STREAM=1)
	// and promises to call bn_ctx_get_stream / bn_sync (both flush) before touching the stream itself.
	ctx->lazy_fold = getenv("BN_NO_LAZY_FOLD") == nullptr && getenv("BN_LAZY_ON_SHARED_STREAM") != nullptr;
	if (!ctx->lazy_fold) ctx->tail_max_n_in = 0;
	return BN_OK;
}

int bn_ctx_get_stream(bn_ctx *ctx, void **hip_stream)
{
	BN_REQUIRE(ctx && hip_stream, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	*hip_stream = (void *)ctx->stream;
	return BN_OK;
}

int bn_sync(bn_ctx *ctx)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_HIP(hipStreamSynchronize(ctx->stream));
	return BN_OK;
}

int bn_prof_begin(bn_ctx *ctx)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	for (auto &r : ctx->prof) {
		ctx->ev_pool.push_back(r.a);
		ctx->ev_pool.push_back(r.b);
	}
	ctx->prof.clear();
	ctx->prof_on = true;
	return BN_OK;
}

int bn_prof_end(bn_ctx *ctx, double *ms_by_class, uint64_t *launches_by_class)
{
	BN_REQUIRE(ctx && ms_by_class && launches_by_class, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	ctx->prof_on = false;
	BN_HIP(hipStreamSynchronize(ctx->stream));
	for (int i = 0; i < BN_PROF_N; i++) {
		ms_by_class[i] = 0;
		launches_by_class[i] = 0;
	}
	for (auto &r : ctx->prof) {
		float ms = 0;
		BN_HIP(hipEventElapsedTime(&ms, r.a, r.b));
		ms_by_class[r.cls] += ms;
		launches_by_class[r.cls] += 1;
		ctx->ev_pool.push_back(r.a);
		ctx->ev_pool.push_back(r.b);
	}
	ctx->prof.clear();
	return BN_OK;
}

int bn_timer_begin(bn_ctx *ctx)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_HIP(hipEventRecord(ctx->ev0, ctx->stream));
	return BN_OK;
}

int bn_timer_end_ms(bn_ctx *ctx, float *ms)
{
	BN_REQUIRE(ctx && ms, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_HIP(hipEventRecord(ctx->ev1, ctx->stream));
	BN_HIP(hipEventSynchronize(ctx->ev1));
	BN_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
	return BN_OK;
}

// ---------------------------------------------------------------------------------- ComputeLayer
int bn_copy_h2d(bn_ctx *ctx, const bn_f128 *h_src, uint64_t src_len, void *d_dst, uint64_t dst_len)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(src_len == dst_len, "precondition: src and dst buffers must have the same length");
	if (src_len == 0) return BN_OK;
	BN_HIP(hipMemcpyAsync(d_dst, h_src, src_len * sizeof(f128), hipMemcpyHostToDevice, ctx->stream));
	// the host buffer is caller-owned pageable memory: make the copy complete before returning
	BN_HIP(hipStreamSynchronize(ctx->stream));
	return BN_OK;
}

int bn_copy_d2h(bn_ctx *ctx, const void *d_src, uint64_t src_len, bn_f128 *h_dst, uint64_t dst_len)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	{
		// a read does not invalidate the host mirror of a tiny fold -- and may create it
		int rc_ = flush_pending(ctx, false, /*publish_tiny=*/ctx->lazy_fold);
		if (rc_) return rc_;
	}
	BN_REQUIRE(src_len == dst_len, "precondition: src and dst buffers must have the same length");
	if (src_len == 0) return BN_OK;
	if (ctx->mirror.valid) {
		for (uint32_t i = 0; i < ctx->mirror.count; i++) {
			const char *base = (const char *)ctx->mirror.ptr[i];
			const char *p = (const char *)d_src;
			if (p >= base && p + src_len * sizeof(f128) <= base + (size_t)ctx->mirror.n * sizeof(f128)) {
				volatile uint64_t *seqw = &ctx->h_mail[64].lo;
				uint64_t spins = 0;
				while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != ctx->mirror.seq) {
					if (++spins > (1ull << 22)) {
						BN_HIP(hipStreamSynchronize(ctx->stream));
						if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != ctx->mirror.seq)
							return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
						break;
					}
				}
				const size_t off = (size_t)i * ctx->mirror.n + (size_t)(p - base) / sizeof(f128);
				for (uint64_t e = 0; e < src_len; e++) {
					h_dst[e].lo = __atomic_load_n(&ctx->h_mail[off + e].lo, __ATOMIC_RELAXED);
					h_dst[e].hi = __atomic_load_n(&ctx->h_mail[off + e].hi, __ATOMIC_RELAXED);
				}
				return BN_OK;
			}
		}
	}
	BN_HIP(hipMemcpyAsync(h_dst, d_src, src_len * sizeof(f128), hipMemcpyDeviceToHost, ctx->stream));
	BN_HIP(hipStreamSynchronize(ctx->stream));
	return BN_OK;
}

int bn_copy_d2d(bn_ctx *ctx, const void *d_src, uint64_t src_len, void *d_dst, uint64_t dst_len)
{
	BN_REQUIRE(ctx, "null ctx");
	BN_ENTER(ctx);
	BN_REQUIRE(src_len == dst_len, "precondition: src and dst buffers must have the same length");
	if (src_len == 0) return BN_OK;
	ctx->mirror.valid = false;
	if (ctx->lazy_fold && !ctx->pend.active && ctx->pend_copies.size() < 8) {
		// deferred: a fold into d_dst may absorb it (see bn_ctx::pending_copy)
		ctx->pend_copies.push_back({d_src, d_dst, src_len});
		return BN_OK;
	}
	BN_FLUSH(ctx);
	BN_HIP(hipMemcpyAsync(d_dst, d_src, src_len * sizeof(f128), hipMemcpyDeviceToDevice, ctx->stream));
	return BN_OK;
}

int bn_fill(bn_ctx *ctx, void *d_dst, uint64_t n, const bn_f128 *value)
{
	BN_REQUIRE(ctx && value, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_HIP(bn::launch_fill(ctx->stream, d_dst, n, to_f(value)));
	return BN_OK;
}

// ---------------------------------------------------------------------------------- expressions
int bn_expr_compile(bn_ctx *ctx, const bn_step *steps, uint64_t n_steps, bn_expr **out)
{
	BN_REQUIRE(ctx && out, "null argument");
	BN_ENTER(ctx);
	BN_REQUIRE(n_steps == 0 || steps, "null steps");
	bn_expr *e = new bn_expr();
	e->device = ctx->device;
	e->steps.assign(steps, steps + n_steps);
	uint32_t n_vars = 0;
	// symbolic pass: which steps are pure products of variables?
	std::vector<std::vector<uint32_t>> prod(n_steps);
	std::vector<bool> is_prod(n_steps, false);
	for (uint64_t s = 0; s < n_steps; s++) {
		const bn_step &st = steps[s];
		switch (st.kind) {
		case BN_STEP_VAR:
			if (st.a + 1 > n_vars) n_vars = st.a + 1;
			is_prod[s] = true;
			prod[s] = {st.a};
			break;
		case BN_STEP_ADD:
		case BN_STEP_MUL:
			if (st.a >= s || st.b >= s) {
				delete e;
				return bn::fail(BN_ERR_INPUT_VALIDATION, "input validation: circuit step refers to a later step");
			}
			if (st.kind == BN_STEP_MUL && is_prod[st.a] && is_prod[st.b]) {
				is_prod[s] = true;
				prod[s] = prod[st.a];
				prod[s].insert(prod[s].end(), prod[st.b].begin(), prod[st.b].end());
			}
			break;
		case BN_STEP_POW:
			if (st.a >= s) {
				delete e;
				return bn::fail(BN_ERR_INPUT_VALIDATION, "input validation: circuit step refers to a later step");
			}
			break;
		case BN_STEP_CONST:
			break;
		default:
			delete e;
			return bn::fail(BN_ERR_INPUT_VALIDATION, "input validation: unknown circuit step kind");
		}
	}
	e->n_vars = n_vars;
	if (n_steps && is_prod[n_steps - 1] && prod[n_steps - 1].size() <= 4) {
		e->shape = bn_expr::PRODUCT;
		e->product_vars = prod[n_steps - 1];
	}
	// the device copy of the steps is only needed by the generic interpreter kernels: uploaded on
	// first use (ensure_d_steps), so compiling a product composition touches no device memory
	*out = e;
	return BN_OK;
}

static int ensure_d_steps(const bn_expr *e)
{
	if (e->d_steps || e->steps.empty()) return BN_OK;
	const size_t bytes = e->steps.size() * sizeof(bn_step);
	hipError_t err = hipMalloc((void **)&e->d_steps, bytes);
	if (err == hipSuccess) err = hipMemcpy(e->d_steps, e->steps.data(), bytes, hipMemcpyHostToDevice);
	if (err != hipSuccess) return bn::hip_fail(err, "bn_expr upload");
	return BN_OK;
}

int bn_expr_free(bn_expr *expr)
{
	if (!expr) return BN_OK;
	if (expr->d_steps) hipFree(expr->d_steps);
	delete expr;
	return BN_OK;
}

int bn_expr_n_vars(const bn_expr *expr, uint32_t *n_vars)
{
	BN_REQUIRE(expr && n_vars, "null argument");
	*n_vars = expr->n_vars;
	return BN_OK;
}

// ---------------------------------------------------------------------------------- executor ops
int bn_extrapolate_line(bn_ctx *ctx, void *d_evals_0, uint64_t n0, const void *d_evals_1, uint64_t n1, const bn_f128 *z)
{
	BN_REQUIRE(ctx && z, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(n0 == n1, "evals_0 and evals_1 must be the same length");
	prof_scope ps(ctx, BN_PROF_FOLD);
	BN_HIP(bn::launch_extrapolate_line(ctx->stream, ctx->n_cu, d_evals_0, d_evals_1, n0, to_f(z)));
	return BN_OK;
}

int bn_extrapolate_line_batch(bn_ctx *ctx, void *const *d_evals_0, const void *const *d_evals_1, uint32_t count, uint64_t n,
                              const bn_f128 *z)
{
	BN_REQUIRE(ctx && z && d_evals_0 && d_evals_1, "null argument");
	BN_ENTER(ctx);
	BN_REQUIRE(count <= (uint32_t)bn::kFoldBatchMax, "too many slices in one extrapolate_line batch");
	if (count == 0) return BN_OK;
	ctx->mirror.valid = false;
	// deferred copies whose destination is one of the evals_0 are absorbed (every one of them must
	// be, otherwise they all run now, in issue order)
	const void *src0[bn::kFoldBatchMax];
	for (uint32_t i = 0; i < count; i++) src0[i] = d_evals_0[i];
	if (!ctx->pend_copies.empty() && !ctx->pend.active) {
		size_t absorbed = 0;
		for (const auto &c : ctx->pend_copies)
			for (uint32_t i = 0; i < count; i++)
				if (c.dst == d_evals_0[i] && c.n == n && src0[i] == d_evals_0[i]) {
					src0[i] = c.src;
					absorbed++;
					break;
				}
		// an absorbed copy is read while the fold writes: its source must not overlap any array the
		// batch writes (e.g. a copy chain s -> d1 -> d2 followed by a fold of both)
		bool clash = false;
		auto overlaps = [&](const void *p, const void *q) {
			const char *a = (const char *)p, *b = (const char *)q;
			return a < b + n * sizeof(f128) && b < a + n * sizeof(f128);
		};
		for (uint32_t i = 0; i < count && !clash; i++)
			if (src0[i] != d_evals_0[i])
				for (uint32_t j = 0; j < count; j++)
					if (overlaps(src0[i], d_evals_0[j])) clash = true;
		if (absorbed == ctx->pend_copies.size() && !clash) {
			ctx->pend_copies.clear();
		} else {
			for (uint32_t i = 0; i < count; i++) src0[i] = d_evals_0[i];
		}
	}
	{
		// a resident tail kernel survives this call only if the batch is the fold it is parked for
		bool keep_tail = false;
		if (ctx->tail.active && count == 2 && 2 * n == ctx->tail.n_in_next) {
			auto continues = [&](uint32_t i, uint32_t j) {
				return d_evals_0[i] == ctx->tail.out[j] && src0[i] == d_evals_0[i] &&
				       (const char *)d_evals_1[i] == (const char *)d_evals_0[i] + n * sizeof(f128);
			};
			keep_tail = (continues(0, 0) && continues(1, 1)) || (continues(0, 1) && continues(1, 0));
		}
		int rc_ = flush_pending(ctx, keep_tail);
		if (rc_) return rc_;
	}
	// Deferred: the next API call launches it -- or, if that call is the round evaluation of exactly
	// these arrays, both run as one kernel (kernels_foldeval9.hip).
	ctx->pend.active = true;
	ctx->pend.count = count;
	ctx->pend.n = n;
	ctx->pend.z = to_f(z);
	for (uint32_t i = 0; i < count; i++) {
		ctx->pend.x0[i] = d_evals_0[i];
		ctx->pend.x1[i] = d_evals_1[i];
		ctx->pend.src0[i] = src0[i];
	}
	if (!ctx->lazy_fold) BN_FLUSH(ctx);
	return BN_OK;
}

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

// XOR of n_groups partial results in d_result[0 .. n_groups) -> host, through the zero-copy mailbox
// (one tiny kernel instead of a device-to-host copy plus a stream synchronisation)
static int publish_result(bn_ctx *ctx, uint32_t n_groups, bn_f128 *h_out)
{
	const uint64_t seq = ++ctx->mail_seq;
	BN_HIP(bn::launch_xor_publish(ctx->stream, ctx->d_result, n_groups, 1, ctx->d_result + 96, ctx->d_mail, seq));
	volatile uint64_t *seqw = &ctx->h_mail[64].lo;
	uint64_t spins = 0;
	while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) {
		if (++spins > (1ull << 22)) {
			BN_HIP(hipStreamSynchronize(ctx->stream));
			if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq)
				return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
			break;
		}
	}
	h_out->lo = __atomic_load_n(&ctx->h_mail[0].lo, __ATOMIC_RELAXED);
	h_out->hi = __atomic_load_n(&ctx->h_mail[0].hi, __ATOMIC_RELAXED);
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
	if (left)
		BN_HIP(bn::launch_fold_left(ctx->stream, ctx->n_cu, d_mat, tower_level, d_vec, vec_len, d_out, out_len));
	else
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

static int upload_s_evals(bn_ctx *ctx, const uint64_t *h_s_evals, uint64_t **d_out, size_t extra_bytes, void **extra)
{
	const size_t sb = sizeof(uint64_t) * BN_NTT_MAX_DIM * BN_NTT_MAX_DIM;
	char *scr = (char *)bn::ctx_scratch(ctx, sb + extra_bytes);
	if (!scr)
		return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (scratch)");
	BN_HIP(hipMemcpyAsync(scr, h_s_evals, sb, hipMemcpyHostToDevice, ctx->stream));
	BN_HIP(hipStreamSynchronize(ctx->stream)); // h_s_evals is caller-owned pageable memory
	*d_out = (uint64_t *)scr;
	if (extra) *extra = scr + sb;
	return BN_OK;
}

int bn_fri_fold(bn_ctx *ctx, const uint64_t *h_s_evals, uint32_t tw_level, uint32_t log_domain, uint32_t log_len,
                uint32_t log_batch_size, const bn_f128 *h_challenges, uint32_t n_challenges, const void *d_in, uint64_t in_len,
                void *d_out, uint64_t out_len)
{
	BN_REQUIRE(ctx && h_s_evals, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
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
	                           d_out, out_len, pp));
	return BN_OK;
}

// rows -> device array of row pointers, staged in the tail of the result mailbox area
static int upload_ptrs(bn_ctx *ctx, const void *const *ptrs, uint32_t n, const void ***d_ptrs)
{
	// use slots [128, 256) of the mailbox: 128 * 16 B = 2 KiB = 256 pointers
	BN_REQUIRE(n <= 256, "too many rows");
	void *dst = (void *)(ctx->d_result + 128);
	BN_HIP(hipMemcpyAsync(dst, ptrs, n * sizeof(void *), hipMemcpyHostToDevice, ctx->stream));
	BN_HIP(hipStreamSynchronize(ctx->stream));
	*d_ptrs = (const void **)dst;
	return BN_OK;
}

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
	const void **d_ptrs = nullptr;
	int rc = upload_ptrs(ctx, d_rows, n_rows, &d_ptrs);
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
	const void *src = d_in;
	for (uint32_t r = 0; r < n_rounds; r++) {
		BN_HIP(bn::launch_mul9(ctx->stream, ctx->n_cu, src, 2, src, 2, 1, d_round_outs[r], round_lens[r]));
		src = d_round_outs[r];
	}
	return BN_OK;
}

// ---------------------------------------------------------------------------------- kernels
static uint32_t map_log_len(const bn_memmap &m) { return m.kind == BN_MAP_LOCAL ? m.log_size : ilog2(m.len); }

int bn_log_chunks_range(const bn_memmap *maps, uint32_t n_maps, uint32_t *start, uint32_t *end)
{
	BN_REQUIRE(maps && n_maps > 0 && start && end, "log_chunks_range needs at least one mapping");
	uint32_t e = ~0u;
	for (uint32_t i = 0; i < n_maps; i++) {
		uint32_t hi;
		if (maps[i].kind == BN_MAP_LOCAL) {
			hi = maps[i].log_size;
		} else {
			BN_REQUIRE(is_pow2(maps[i].len), "mapped buffer length must be a power of two");
			uint32_t log_data = ilog2(maps[i].len);
			uint32_t lm = maps[i].log_min_chunk_size; // max(.., log2 ALIGNMENT = 0)
			if (lm > log_data) lm = log_data;
			hi = log_data - lm;
		}
		if (hi < e) e = hi;
	}
	*start = 0;
	*end = e;
	return BN_OK;
}

int bn_pick_log_chunks(const bn_memmap *maps, uint32_t n_maps, uint32_t *log_chunks)
{
	uint32_t s, e;
	int rc = bn_log_chunks_range(maps, n_maps, &s, &e);
	if (rc) return rc;
	// One logical chunk: the grid itself is the parallel decomposition and the cross-workgroup
	// XOR reduction is done on the device, so the closure is recorded once over whole buffers.
	*log_chunks = s;
	return BN_OK;
}

namespace {
// how a kernel-buffer slice is realised on the device
struct slice_view {
	const char *p = nullptr; // direct data
	const char *q = nullptr; // if non-null: value = p ^ q (a Local buffer defined by ADD and not materialised)
	bool zero = false;       // untouched Local buffer
	uint64_t len = 0;
};
} // namespace

int bn_kernel_launch(bn_ctx *ctx, const bn_memmap *maps, uint32_t n_maps, const bn_kop *ops, uint32_t n_ops,
                     const uint32_t *ret_values, uint32_t n_ret, uint32_t log_chunks, bn_f128 *h_out, void *d_out)
{
	BN_REQUIRE(ctx && maps && n_maps > 0, "kernel launch needs at least one mapping");
	BN_ENTER(ctx);
	uint32_t lo_c, hi_c;
	int rc = bn_log_chunks_range(maps, n_maps, &lo_c, &hi_c);
	if (rc) return rc;
	BN_REQUIRE(log_chunks == 0, "this backend records kernels with log_chunks = bn_pick_log_chunks() = 0");
	BN_REQUIRE(n_ret <= 64, "too many returned values");
	hipStream_t s = ctx->stream;

	// A deferred fold survives into this launch only if the kernel has the calculate_round_evals
	// shape (two bivariate-product sums, Local "lo + hi" operands, nothing written to memory); the
	// launch site below then checks that it reads exactly the folded arrays.
	if (!ctx->pend.active) BN_FLUSH(ctx); // (deferred copies; a parked tail kernel without a fold to run)
	if (ctx->pend.active) {
		uint32_t n_sum = 0;
		bool pure = n_ret > 0 && ctx->pend.count == 2;
		for (uint32_t o = 0; o < n_ops && pure; o++) {
			const bn_kop &op = ops[o];
			if (op.kind == BN_KOP_SUM_COMPOSITION) {
				n_sum++;
				if (!op.expr || op.expr->shape != bn_expr::PRODUCT || op.expr->product_vars.size() != 2) pure = false;
			} else if (op.kind == BN_KOP_ADD) {
				if (op.dst.buf >= n_maps || maps[op.dst.buf].kind != BN_MAP_LOCAL) pure = false;
			} else if (op.kind != BN_KOP_DECL_VALUE) {
				pure = false;
			}
		}
		if (!pure || n_sum != 2) BN_FLUSH(ctx);
	}

	// Local buffers are virtual until something forces them into memory.
	struct local_state {
		bool defined = false;           // written by an ADD covering the whole buffer
		const char *p = nullptr, *q = nullptr;
		char *mem = nullptr;            // materialised storage
	};
	std::vector<local_state> loc(n_maps);
	std::vector<uint64_t> buf_len(n_maps);
	size_t local_bytes = 0;
	for (uint32_t i = 0; i < n_maps; i++) {
		buf_len[i] = maps[i].kind == BN_MAP_LOCAL ? ((uint64_t)1 << maps[i].log_size) : maps[i].len;
		if (maps[i].kind == BN_MAP_LOCAL) local_bytes += buf_len[i] * sizeof(f128);
	}
	(void)map_log_len;

	// Do we need real memory for Local buffers?  Only if a Local is read/written in a way the
	// virtual form cannot express (partial slices, ADD_ASSIGN into it, ADD of virtual operands).
	bool need_materialise = false;
	for (uint32_t o = 0; o < n_ops && !need_materialise; o++) {
		const bn_kop &op = ops[o];
		auto whole = [&](const bn_kslice &sl) { return sl.off == 0 && sl.len == buf_len[sl.buf]; };
		if (op.kind == BN_KOP_ADD) {
			BN_REQUIRE(op.dst.buf < n_maps && op.src1.buf < n_maps && op.src2.buf < n_maps, "slice refers to an unknown buffer");
			if (maps[op.dst.buf].kind == BN_MAP_LOCAL) {
				if (!whole(op.dst) || maps[op.src1.buf].kind == BN_MAP_LOCAL || maps[op.src2.buf].kind == BN_MAP_LOCAL)
					need_materialise = true;
			}
		} else if (op.kind == BN_KOP_ADD_ASSIGN) {
			BN_REQUIRE(op.dst.buf < n_maps && op.src1.buf < n_maps, "slice refers to an unknown buffer");
			if (maps[op.dst.buf].kind == BN_MAP_LOCAL || maps[op.src1.buf].kind == BN_MAP_LOCAL)
				need_materialise = true;
		} else if (op.kind == BN_KOP_SUM_COMPOSITION) {
			BN_REQUIRE(op.expr, "sum_composition_evals without a compiled expression");
			for (uint32_t r = 0; r < op.n_rows; r++) {
				BN_REQUIRE(op.rows[r].buf < n_maps, "slice refers to an unknown buffer");
				if (maps[op.rows[r].buf].kind == BN_MAP_LOCAL && !whole(op.rows[r]))
					need_materialise = true;
			}
			if (op.expr->shape != bn_expr::PRODUCT)
				for (uint32_t r = 0; r < op.n_rows; r++)
					if (maps[op.rows[r].buf].kind == BN_MAP_LOCAL)
						need_materialise = true;
		}
	}
	if (need_materialise && local_bytes) {
		char *mem = (char *)bn::ctx_scratch(ctx, local_bytes);
		if (!mem)
			return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (Local kernel buffers)");
		BN_HIP(hipMemsetAsync(mem, 0, local_bytes, s)); // "initialized with zeros", layer.rs:154-156
		size_t off = 0;
		for (uint32_t i = 0; i < n_maps; i++)
			if (maps[i].kind == BN_MAP_LOCAL) {
				loc[i].mem = mem + off;
				off += buf_len[i] * sizeof(f128);
			}
	}

	auto view = [&](const bn_kslice &sl) -> slice_view {
		slice_view v;
		v.len = sl.len;
		const bn_memmap &m = maps[sl.buf];
		if (m.kind != BN_MAP_LOCAL) {
			v.p = (const char *)m.d_data + sl.off * sizeof(f128);
		} else if (loc[sl.buf].mem) {
			v.p = loc[sl.buf].mem + sl.off * sizeof(f128);
		} else if (loc[sl.buf].defined) {
			v.p = loc[sl.buf].p;
			v.q = loc[sl.buf].q;
		} else {
			v.zero = true;
		}
		return v;
	};

	// device accumulators: S slots in the mailbox [0, 64), values in [64, 128)
	uint32_t n_values = 0;
	for (uint32_t o = 0; o < n_ops; o++)
		if (ops[o].kind == BN_KOP_DECL_VALUE && ops[o].value + 1 > n_values) n_values = ops[o].value + 1;
	BN_REQUIRE(n_values <= (uint32_t)bn::kFinMaxValues, "too many kernel values");
	for (uint32_t i = 0; i < n_ret; i++)
		BN_REQUIRE(ret_values[i] < n_values, "returned value was never declared");
	std::vector<f128> h_values(n_values ? n_values : 1, bn::f128_zero());
	std::vector<bn::fin_term> terms;
	uint32_t n_slots = 0;
	bool finalized_in_kernel = false;
	uint64_t fused_seq = 0;
	f128 *d_S = ctx->d_result;         // [0,64)
	f128 *d_rets = ctx->d_result + 96;  // [96,128)
	bool has_sum = false;
	for (uint32_t o = 0; o < n_ops; o++) has_sum |= ops[o].kind == BN_KOP_SUM_COMPOSITION;
	if (!ctx->s_clean && has_sum) {
		BN_HIP(hipMemsetAsync(d_S, 0, 64 * sizeof(f128), s));
		ctx->s_clean = true;
	}
	const bool was_clean_or_zeroed = ctx->s_clean;
	ctx->s_clean = false; // until the finalize kernel of THIS call has re-zeroed the slots it used

	for (uint32_t o = 0; o < n_ops; o++) {
		const bn_kop &op = ops[o];
		switch (op.kind) {
		case BN_KOP_DECL_VALUE:
			h_values[op.value] = f128{op.scalar.lo, op.scalar.hi};
			break;
		case BN_KOP_ADD: {
			BN_REQUIRE(maps[op.dst.buf].kind != BN_MAP_CHUNKED, "add: destination buffer is read-only");
			BN_REQUIRE(op.src1.len == op.dst.len && op.src2.len == op.dst.len, "add: slice lengths differ");
			BN_REQUIRE(op.dst.off + op.dst.len <= buf_len[op.dst.buf] && op.src1.off + op.src1.len <= buf_len[op.src1.buf] &&
			               op.src2.off + op.src2.len <= buf_len[op.src2.buf],
			           "add: slice out of range");
			if (maps[op.dst.buf].kind == BN_MAP_LOCAL && !loc[op.dst.buf].mem) {
				// virtual definition: dst := src1 ^ src2 (never touches HBM)
				slice_view a = view(op.src1), b = view(op.src2);
				loc[op.dst.buf].defined = true;
				loc[op.dst.buf].p = a.p;
				loc[op.dst.buf].q = b.p;
			} else {
				slice_view a = view(op.src1), b = view(op.src2), d = view(op.dst);
				BN_REQUIRE(!a.q && !b.q && !a.zero && !b.zero, "add: unsupported operand form");
				BN_HIP(bn::launch_add(s, (void *)d.p, a.p, b.p, op.dst.len));
			}
			break;
		}
		case BN_KOP_ADD_ASSIGN: {
			BN_REQUIRE(maps[op.dst.buf].kind != BN_MAP_CHUNKED, "add_assign: destination buffer is read-only");
			BN_REQUIRE(op.src1.len == op.dst.len, "add_assign: slice lengths differ");
			BN_REQUIRE(op.dst.off + op.dst.len <= buf_len[op.dst.buf] && op.src1.off + op.src1.len <= buf_len[op.src1.buf],
			           "add_assign: slice out of range");
			slice_view a = view(op.src1), d = view(op.dst);
			BN_REQUIRE(!a.q && !a.zero && !d.q && !d.zero, "add_assign: unsupported operand form");
			BN_HIP(bn::launch_add_assign(s, (void *)d.p, a.p, op.dst.len));
			break;
		}
		case BN_KOP_SUM_COMPOSITION: {
			BN_REQUIRE(op.value < n_values, "sum_composition_evals: accumulator was never declared");
			BN_REQUIRE(op.n_rows >= op.expr->n_vars, "composition does not match the number of input rows");
			BN_REQUIRE(op.expr->steps.size() <= 64, "circuit too large for this backend (max 64 steps)");
			const uint64_t row_len = op.n_rows ? op.rows[0].len : 0;
			for (uint32_t r = 0; r < op.n_rows; r++) {
				BN_REQUIRE(op.rows[r].len == row_len, "sum_composition_evals: rows differ in length");
				BN_REQUIRE(op.rows[r].off + op.rows[r].len <= buf_len[op.rows[r].buf], "sum_composition_evals: slice out of range");
			}
			BN_REQUIRE(n_slots + 2 <= 64, "too many sum_composition_evals in one kernel");
			const uint32_t slot = n_slots;
			if (op.expr->shape == bn_expr::PRODUCT) {
				// fused pairing: if the NEXT sum op uses the same expression and its factors are the
				// "infinity" versions (Local = lo + hi with hi == this op's row) of this op's factors,
				// do both with one pass over the data.
				const uint32_t k = (uint32_t)op.expr->product_vars.size();
				const void *hi[4] = {nullptr, nullptr, nullptr, nullptr}, *lo[4] = {nullptr, nullptr, nullptr, nullptr};
				bool direct = true;
				std::vector<slice_view> fv(k);
				for (uint32_t j = 0; j < k; j++) {
					fv[j] = view(op.rows[op.expr->product_vars[j]]);
					if (fv[j].q || fv[j].zero) direct = false;
				}
				// look ahead for the partner op (skipping ADD ops that define Locals and DECLs)
				int partner = -1;
				if (direct) {
					for (uint32_t o2 = o + 1; o2 < n_ops; o2++) {
						if (ops[o2].kind == BN_KOP_SUM_COMPOSITION) {
							if (ops[o2].expr == op.expr && ops[o2].n_rows == op.n_rows) partner = (int)o2;
							break;
						}
						if (ops[o2].kind == BN_KOP_ADD_ASSIGN) break;
					}
				}
				bool fused = false;
				if (partner >= 0) {
					// evaluate the intervening ADD / DECL ops now (they only define virtual Locals)
					bool ok = true;
					for (uint32_t o2 = o + 1; o2 < (uint32_t)partner && ok; o2++) {
						const bn_kop &mid = ops[o2];
						if (mid.kind == BN_KOP_DECL_VALUE) continue;
						if (mid.kind != BN_KOP_ADD || maps[mid.dst.buf].kind != BN_MAP_LOCAL || loc[mid.dst.buf].mem) ok = false;
					}
					if (ok) {
						// tentatively compute partner views
						std::vector<local_state> saved = loc;
						for (uint32_t o2 = o + 1; o2 < (uint32_t)partner; o2++) {
							const bn_kop &mid = ops[o2];
							if (mid.kind != BN_KOP_ADD) continue;
							slice_view a = view(mid.src1), b = view(mid.src2);
							if (a.q || b.q || a.zero || b.zero) { ok = false; break; }
							loc[mid.dst.buf].defined = true;
							loc[mid.dst.buf].p = a.p;
							loc[mid.dst.buf].q = b.p;
						}
						const bn_kop &pop = ops[partner];
						for (uint32_t j = 0; j < k && ok; j++) {
							slice_view pv = view(pop.rows[op.expr->product_vars[j]]);
							if (pv.len != row_len || pv.zero) { ok = false; break; }
							hi[j] = fv[j].p;
							if (!pv.q && pv.p == fv[j].p) {
								lo[j] = nullptr; // same factor at both points
							} else if (pv.q && pv.q == fv[j].p) {
								lo[j] = pv.p;    // Local = lo + hi
							} else if (pv.q && pv.p == fv[j].p) {
								lo[j] = pv.q;
							} else {
								ok = false;
							}
						}
						if (ok) {
							BN_REQUIRE(n_slots + 2 <= 64, "too many sum_composition_evals in one kernel");
							const bn_kop &pop2 = ops[partner];
							BN_REQUIRE(pop2.value < n_values, "sum_composition_evals: accumulator was never declared");
							// DECLs between the two ops
							for (uint32_t o2 = o + 1; o2 < (uint32_t)partner; o2++)
								if (ops[o2].kind == BN_KOP_DECL_VALUE)
									h_values[ops[o2].value] = f128{ops[o2].scalar.lo, ops[o2].scalar.hi};
							terms.push_back(bn::fin_term{op.value, slot, f128{op.scalar.lo, op.scalar.hi}});
							terms.push_back(bn::fin_term{pop2.value, slot + 1, f128{pop2.scalar.lo, pop2.scalar.hi}});
							// If this pair is the whole kernel (the calculate_round_evals shape), the finalize
							// step rides in the same launch: the last workgroup folds and publishes the values.
							bool in_kernel = false;
							if ((uint32_t)partner + 1 == n_ops && n_slots == 0 && n_ret > 0 && n_ret <= (uint32_t)bn::kFinMaxRets &&
							    n_values <= (uint32_t)bn::kFinMaxValues) {
								bn::fin_fuse fz{};
								fz.args.n_terms = 2;
								fz.args.n_values = n_values;
								fz.args.n_ret = n_ret;
								fz.args.n_slots = 2;
								fz.args.seq = h_out ? ++ctx->mail_seq : 0;
								fz.args.terms[0] = terms[terms.size() - 2];
								fz.args.terms[1] = terms[terms.size() - 1];
								for (uint32_t v = 0; v < n_values; v++) fz.args.init[v] = h_values[v];
								for (uint32_t r = 0; r < n_ret; r++) fz.args.ret_ids[r] = ret_values[r];
								fz.S = d_S;
								fz.rets = d_out ? (f128 *)d_out : d_rets;
								fz.mail = ctx->d_mail;
								fz.counter = ctx->d_ticket;
								hipError_t fe = hipErrorNotSupported;
								if (ctx->pend.active) {
									// fold + evaluate in one pass: this launch reads the halves of exactly the two
									// arrays the deferred fold writes (evals_1 directly behind evals_0, in place)
									const bn_ctx::pending_fold &pf = ctx->pend;
									auto reads_folded = [&](uint32_t j, uint32_t i) {
										return lo[j] == pf.x0[i] && (const char *)hi[j] == (const char *)lo[j] + row_len * sizeof(f128);
									};
									int perm = -1;
									if (k == 2 && pf.n == 2 * row_len && pf.x0[0] != pf.x0[1] && lo[0] && lo[1]) {
										if (reads_folded(0, 0) && reads_folded(1, 1)) perm = 0;
										else if (reads_folded(0, 1) && reads_folded(1, 0)) perm = 1;
									}
									if (perm >= 0) {
										bn::foldeval_args fa{};
										for (uint32_t j = 0; j < 2; j++) {
											const uint32_t i = perm ? 1 - j : j;
											fa.x0[j] = pf.src0[i];
											fa.x1[j] = pf.x1[i];
											fa.out[j] = pf.x0[i];
										}
										const uint64_t n_in = 2 * pf.n;
										// (a) a resident tail kernel is parked for exactly this round: hand it z
										if (ctx->tail.active) {
											bn_ctx::tail_state &tl = ctx->tail;
											const bool same = h_out && !d_out && n_in == tl.n_in_next && fa.x0[0] == fa.out[0] && fa.x0[1] == fa.out[1] &&
											                  ((fa.out[0] == tl.out[0] && fa.out[1] == tl.out[1]) || (fa.out[0] == tl.out[1] && fa.out[1] == tl.out[0])) &&
											                  fz.args.seq == tl.seq0 + tl.round + 1 && recipe_bytes(fz.args) == tl.recipe &&
											                  __atomic_load_n(tail_status(ctx), __ATOMIC_ACQUIRE) != tl.id;
											if (same) {
												tl.round++;
												ctx->h_mail[81].lo = pf.z.lo;
												ctx->h_mail[81].hi = pf.z.hi;
												__atomic_store_n(tail_cmd(ctx), (tl.id << 20) | tl.round, __ATOMIC_RELEASE);
												volatile uint64_t *seqw = &ctx->h_mail[64].lo;
												bool got = false;
												for (uint64_t spins = 0;; spins++) {
													if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) == fz.args.seq) { got = true; break; }
													if (__atomic_load_n(tail_status(ctx), __ATOMIC_ACQUIRE) == tl.id) {
														// the kernel left (bounded spin ran out) -- did it answer first?
														got = __atomic_load_n(seqw, __ATOMIC_ACQUIRE) == fz.args.seq;
														break;
													}
													if (spins > (1ull << 26)) {
														// neither word moves: the kernel faulted or the device hangs.  Same fallback as
														// the other mailbox waits: let the stream report it.
														tl.active = false;
														BN_HIP(hipStreamSynchronize(s));
														return bn::fail(BN_ERR_DEVICE, "device error: resident tail kernel stopped answering");
													}
												}
												if (got) {
													for (uint32_t r = 0; r < n_ret; r++) {
														h_out[r].lo = __atomic_load_n(&ctx->h_mail[r].lo, __ATOMIC_RELAXED);
														h_out[r].hi = __atomic_load_n(&ctx->h_mail[r].hi, __ATOMIC_RELAXED);
													}
													ctx->pend.active = false;
													tl.n_in_next = n_in >> 1;
													if (n_in <= 4) tl.active = false; // it has just run its last round and exits
													ctx->s_clean = true;
													return BN_OK;
												}
												tl.active = false; // gone without doing this round: run it the normal way
												BN_HIP(hipStreamSynchronize(s));
											} else {
												rc = tail_cancel(ctx);
												if (rc) return rc;
											}
										}
										// (b) small arrays: start a resident tail kernel with this round
										if (fe == hipErrorNotSupported && h_out && !d_out && ctx->tail_max_n_in && n_in <= ctx->tail_max_n_in && n_in >= 8) {
											bn_ctx::tail_state &tl = ctx->tail;
											const uint64_t id = ++ctx->tail_counter;
											prof_scope ps(ctx, BN_PROF_TAIL);
											fe = bn::launch_foldeval_tail(s, fa, n_in, pf.z, d_S + slot, fz, (const uint64_t *)&ctx->d_mail[80].lo,
											                              (uint64_t *)&ctx->d_mail[82].lo, id);
											if (fe == hipSuccess) {
												tl.active = true;
												tl.id = id;
												tl.round = 0;
												tl.n_in_next = n_in >> 1;
												tl.out[0] = fa.out[0];
												tl.out[1] = fa.out[1];
												tl.seq0 = fz.args.seq;
												tl.recipe = recipe_bytes(fz.args);
												ctx->pend.active = false;
											}
										}
										// (c) one fused kernel for this round
										if (fe == hipErrorNotSupported) {
											const bool mfma = bn::mfma_applies(ctx->n_cu, n_in >> 2);
											prof_scope ps(ctx, mfma ? BN_PROF_FOLD_EVAL_MFMA : (bn::foldeval9_is_small(ctx->n_cu, n_in) ? BN_PROF_FOLD_EVAL_SMALL : BN_PROF_FOLD_EVAL));
											fe = mfma ? bn::launch_foldeval_mfma(s, ctx->n_cu, fa, n_in, pf.z, d_S + slot, &fz)
											          : bn::launch_foldeval9(s, ctx->n_cu, fa, n_in, pf.z, d_S + slot, &fz);
											if (fe == hipSuccess) ctx->pend.active = false;
										}
									} else if (ctx->tail.active) {
										rc = tail_cancel(ctx);
										if (rc) return rc;
									}
									if (ctx->pend.active) BN_FLUSH(ctx);
								}
								if (fe == hipErrorNotSupported) {
									prof_scope ps(ctx, k == 2 && bn::mfma_applies(ctx->n_cu, row_len) ? BN_PROF_ROUND_EVAL_MFMA : BN_PROF_ROUND_EVAL);
									fe = bn::launch_roundeval_product(s, ctx->n_cu, hi, lo, k, row_len, d_S + slot, &fz);
								}
								if (fe == hipSuccess) {
									in_kernel = true;
									finalized_in_kernel = true;
									fused_seq = fz.args.seq;
								} else if (fe != hipErrorNotSupported) {
									return bn::hip_fail(fe, "launch_roundeval_product (fused finalize)");
								} else if (h_out) {
									--ctx->mail_seq;
								}
							}
							if (!in_kernel) {
								BN_FLUSH(ctx);
								prof_scope ps(ctx, BN_PROF_ROUND_EVAL);
								BN_HIP(bn::launch_roundeval_product(s, ctx->n_cu, hi, lo, k, row_len, d_S + slot, nullptr));
							}
							n_slots += 2;
							o = (uint32_t)partner; // consumed
							fused = true;
						} else {
							loc = saved;
						}
					}
				}
				if (!fused) {
					// single job: factors may be direct or virtual (p ^ q)
					bool any_virtual = false;
					for (uint32_t j = 0; j < k; j++)
						if (fv[j].q) any_virtual = true;
					bool any_zero = false;
					for (uint32_t j = 0; j < k; j++)
						if (fv[j].zero) any_zero = true;
					if (any_zero || row_len == 0) {
						// a factor is identically zero: contributes nothing
					} else if (!any_virtual) {
						BN_FLUSH(ctx);
						const void *rows[4];
						for (uint32_t j = 0; j < k; j++) rows[j] = fv[j].p;
						BN_HIP(bn::launch_sum_product(s, ctx->n_cu, rows, k, row_len, d_S + slot));
						terms.push_back(bn::fin_term{op.value, slot, f128{op.scalar.lo, op.scalar.hi}});
						terms.push_back(bn::fin_term{op.value, slot + 1, f128{op.scalar.lo, op.scalar.hi}});
					} else {
						// "infinity" job alone: low group = p, high group = p ^ q; only the high sum is wanted
						BN_FLUSH(ctx);
						for (uint32_t j = 0; j < k; j++) {
							hi[j] = fv[j].p;
							lo[j] = fv[j].q; // nullptr => same at both
						}
						BN_HIP(bn::launch_roundeval_product(s, ctx->n_cu, hi, lo, k, row_len, d_S + slot, nullptr));
						terms.push_back(bn::fin_term{op.value, slot + 1, f128{op.scalar.lo, op.scalar.hi}});
					}
					n_slots += 2;
				}
			} else {
				// generic circuit: interpreter over materialised rows
				std::vector<const void *> rows(op.n_rows);
				for (uint32_t r = 0; r < op.n_rows; r++) {
					slice_view v = view(op.rows[r]);
					BN_REQUIRE(!v.q && !v.zero, "generic composition over an unmaterialised Local buffer");
					rows[r] = v.p;
				}
				const void **d_ptrs = nullptr;
				rc = upload_ptrs(ctx, rows.data(), op.n_rows, &d_ptrs);
				if (rc) return rc;
				rc = ensure_d_steps(op.expr);
				if (rc) return rc;
				BN_HIP(bn::launch_sum_composition_generic(s, ctx->n_cu, d_ptrs, op.n_rows, row_len, op.expr->d_steps,
				                                          (uint32_t)op.expr->steps.size(), d_S + slot));
				// the pointer table is reused by the next generic op: keep the stream ordered
				BN_HIP(hipStreamSynchronize(s));
				terms.push_back(bn::fin_term{op.value, slot, f128{op.scalar.lo, op.scalar.hi}});
				n_slots += 2;
			}
			break;
		}
		default:
			return bn::fail(BN_ERR_INPUT_VALIDATION, "input validation: unknown kernel op");
		}
	}

	rc = flush_pending(ctx, /*keep_tail=*/true); // (a launch that ended up reading nothing)
	if (rc) return rc;
	if (n_ret == 0) {
		if (n_slots == 0) ctx->s_clean = was_clean_or_zeroed; // no accumulator was touched by this launch
		return BN_OK;
	}

	// finalize on device: values = init ^ sum coeff*S ; rets gathered into d_rets (and d_out).
	// Everything the kernel needs travels as a by-value kernel argument (no staging copies).
	BN_REQUIRE(terms.size() <= (size_t)bn::kFinMaxTerms, "kernel has too many sum_composition_evals terms");
	BN_REQUIRE(n_values <= (uint32_t)bn::kFinMaxValues, "too many kernel values");
	BN_REQUIRE(n_ret <= (uint32_t)bn::kFinMaxRets, "too many returned values");
	bn::fin_args fa{};
	fa.n_terms = (uint32_t)terms.size();
	fa.n_values = n_values;
	fa.n_ret = n_ret;
	for (size_t t = 0; t < terms.size(); t++) fa.terms[t] = terms[t];
	for (uint32_t v = 0; v < n_values; v++) fa.init[v] = h_values[v];
	for (uint32_t r = 0; r < n_ret; r++) fa.ret_ids[r] = ret_values[r];
	fa.n_slots = n_slots;
	fa.seq = finalized_in_kernel ? fused_seq : (h_out ? ++ctx->mail_seq : 0);
	f128 *rets = d_out ? (f128 *)d_out : d_rets;
	if (!finalized_in_kernel)
		BN_HIP(bn::launch_finalize(s, fa, d_S, rets, ctx->d_mail));
	ctx->s_clean = true; // stream-ordered: the next launch on this stream sees zeroed slots
	if (h_out) {
		// spin on the sequence word the kernel publishes after the values (fine-grained host memory)
		volatile uint64_t *seqw = &ctx->h_mail[64].lo;
		const uint64_t want = fa.seq;
		uint64_t spins = 0;
		while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != want) {
			if (++spins > (1ull << 22)) {
				// not there yet: fall back to a stream sync so device errors surface instead of hanging
				BN_HIP(hipStreamSynchronize(s));
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != want)
					return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
				break;
			}
		}
		for (uint32_t r = 0; r < n_ret; r++) {
			h_out[r].lo = __atomic_load_n(&ctx->h_mail[r].lo, __ATOMIC_RELAXED);
			h_out[r].hi = __atomic_load_n(&ctx->h_mail[r].hi, __ATOMIC_RELAXED);
		}
	}
	return BN_OK;
}

// A small region of fine-grained pinned host memory that the device can read directly (32
// elements): inputs of a few elements can be handed to kernels without an upload.  Not part of the
// reference interface (used for the residual instance of the sharded prover).
int bn_host_scratch(bn_ctx *ctx, void **h_ptr, void **d_ptr, uint64_t *elems)
{
	BN_REQUIRE(ctx && h_ptr && d_ptr && elems, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	*h_ptr = ctx->h_mail + 96;
	*d_ptr = ctx->d_mail + 96;
	*elems = 32;
	return BN_OK;
}

// XOR of n_groups device vectors of group_len (<= 64) elements, returned to the host through the
// zero-copy mailbox.  Not part of the reference interface: it is the combine step behind the
// per-round all_gather of the sharded prover (binius_amd/host/host_capi.cpp).
int bn_xor_reduce(bn_ctx *ctx, const void *d_vals, uint32_t n_groups, uint32_t group_len, bn_f128 *h_out)
{
	BN_REQUIRE(ctx && d_vals && h_out, "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(group_len >= 1 && group_len <= 64 && n_groups >= 1, "xor_reduce: group_len must be in 1..64");
	const uint64_t seq = ++ctx->mail_seq;
	BN_HIP(bn::launch_xor_publish(ctx->stream, (const f128 *)d_vals, n_groups, group_len, ctx->d_result + 96, ctx->d_mail, seq));
	volatile uint64_t *seqw = &ctx->h_mail[64].lo;
	uint64_t spins = 0;
	while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) {
		if (++spins > (1ull << 22)) {
			BN_HIP(hipStreamSynchronize(ctx->stream));
			if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq)
				return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
			break;
		}
	}
	for (uint32_t r = 0; r < group_len; r++) {
		h_out[r].lo = __atomic_load_n(&ctx->h_mail[r].lo, __ATOMIC_RELAXED);
		h_out[r].hi = __atomic_load_n(&ctx->h_mail[r].hi, __ATOMIC_RELAXED);
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
	BN_FLUSH(ctx);
	BN_REQUIRE(batch_size != 0 && n_elems % batch_size == 0, "IncorrectBatchSize");
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
	BN_FLUSH(ctx);
	if (n_items == 0 || item_elems == 0) return BN_OK;
	BN_REQUIRE(n_items <= (1ull << 24) && item_elems <= (1ull << 24) && n_items * item_elems <= (1ull << 26), "gather: too many elements for one call");
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

// ---------------------------------------------------------------------------------- host scalars
int bn_scalar_mul(const bn_f128 *a, const bn_f128 *b, bn_f128 *out)
{
	BN_REQUIRE(a && b && out, "null argument");
	f128 r = bn::mul_slow(to_f(a), to_f(b));
	out->lo = r.lo;
	out->hi = r.hi;
	return BN_OK;
}

int bn_scalar_invert(const bn_f128 *a, bn_f128 *out)
{
	BN_REQUIRE(a && out, "null argument");
	// invert_or_zero semantics (0 -> 0); tower descent, ~1 us on the host
	f128 r = bn::invert_tower(to_f(a));
	if (a->lo == 0 && a->hi == 0) r = bn::f128_zero();
	out->lo = r.lo;
	out->hi = r.hi;
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
	if (elem_level >= 5 && tw_level == 5 && log_y >= 14 && !getenv("BN_NTT_NO_BITSLICE")) {
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
	if (!getenv("BN_NTT_PER_LAYER")) {
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
