// binius_amd/csrc/abi_kernels.cpp -- the recorded-kernel dispatcher behind accumulate_kernels / map_kernels
// (bn_kernel_launch): validates the op list, recognises the round-evaluation shapes, decides between the
// fused fold + evaluation kernels (matrix-core or 9-lane), the resident tail and the generic forms, and
// finalizes on the device.  Also log_chunks_range / pick_log_chunks and the device XOR of gathered partials.
#include "abi_common.hpp"
#include "arm.hpp"
#include "hostmul.hpp"

#include <chrono>

extern "C" {

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
// a * b * eq at prover sizes (the MLE-check round evaluation an unchanged BivariateMLEcheckProver records,
// v3/bivariate_mlecheck.rs:391-520).  The three-factor 9-lane kernel pays two chained bit-sliced products per point
// (0.11 of the roofline); the product is linear in b, so two element-wise passes fold the indicator into b --
// t1 = b_hi * eq, t2 = b_lo * eq, (b_lo + b_hi) * eq = t1 + t2 -- and the bivariate matrix-core kernel does the rest:
// S_1 = sum a_hi * t1, S_inf = sum (a_lo + a_hi) * (t2 + t1).  2 * n elements of context scratch (not while Local
// buffers live there); BN_EQ_ROUTE=0 keeps the three-factor kernel.
hipError_t roundeval_product_routed(bn_ctx *ctx, bool scratch_free, const void *const *hi, const void *const *lo, uint32_t k, uint64_t n,
                                    f128 *d_out, const bn::fin_fuse *fuse)
{
	static const bool route = [] {
		const char *e = bn::settled_knob("BN_EQ_ROUTE");
		return !(e && e[0] == '0');
	}();
	if (route && scratch_free && k == 3 && n >= (1ull << 20) && bn::mfma_applies(ctx->n_cu, n)) {
		int same = -1, n_same = 0;
		for (int j = 0; j < 3; j++)
			if (!lo[j]) {
				same = j;
				n_same++;
			}
		if (n_same == 1) {
			const int x = (same + 1) % 3, y = (same + 2) % 3;
			char *t = (char *)bn::ctx_scratch(ctx, 2 * n * sizeof(f128));
			if (t) {
				void *t1 = t, *t2 = t + n * sizeof(f128);
				hipError_t e = bn::launch_mul9(ctx->stream, ctx->n_cu, hi[y], 1, hi[same], 1, 0, t1, n);
				if (e != hipSuccess) return e;
				e = bn::launch_mul9(ctx->stream, ctx->n_cu, lo[y], 1, hi[same], 1, 0, t2, n);
				if (e != hipSuccess) return e;
				return bn::launch_roundeval_mfma_pair(ctx->stream, ctx->n_cu, hi[x], lo[x], t1, t2, n, d_out, fuse);
			}
		}
	}
	return bn::launch_roundeval_product(ctx->stream, ctx->n_cu, hi, lo, k, n, d_out, fuse);
}

constexpr uint64_t kArmMaxIn = 1ull << 19;     // largest round (elements per array before the fold) that is armed ...
// ... when it runs on the matrix-core kernel.  Round 3 stopped at 2^21 (a launch is a tenth of a 60 us kernel); a launch is still
// ~7 us of dead time against a ~3 us signal however long the kernel is, and a shard of an eight-way split has three more such
// rounds (r = 22 ... 24), so the limit is 2^24 now.  Above that a launch is < 2 % of the kernel, and an armed kernel's wait for
// its challenge would be booked to the kernel by every profiler (rocprofv3 durations of the launches that carry the roofline
// figure).  BN_ARM_MAX_LOG2 moves the limit.  The host's waits cancel an armed kernel before they ever fall back to a stream
// synchronisation, so a long kernel in front is harmless.
uint64_t arm_max_in_mfma()
{
	static const uint64_t v = [] {
		const char *e = bn::settled_knob("BN_ARM_MAX_LOG2");
		const int l = e ? atoi(e) : 24;
		return (uint64_t)1 << (l < 4 ? 4 : (l > 40 ? 40 : l));
	}();
	return v;
}
// ---- two rounds per launch (kernels_foldeval8.hip) ------------------------------------------------------------------
// largest Y (elements per array after the folds of the launch): one 64-point workgroup per CU, 256 CUs
// (BN_TWO_ROUND_MAX_LOG2, 2 .. 20, moves the limit: measurement knob)
uint64_t two_round_max_m()
{
	static const uint64_t v = [] {
		const char *e = bn::settled_knob("BN_TWO_ROUND_MAX_LOG2");
		const int l = e ? atoi(e) : 16;
		return (uint64_t)1 << (l < 2 ? 2 : (l > 20 ? 20 : l));
	}();
	return v;
}
bool two_round_size_ok(uint64_t m) { return m >= 4 && (m & 3) == 0 && m <= two_round_max_m(); }
// the caller's request must be the plain pair (y_1, y_inf) with ONE batch coefficient (the precomputed sums of the next
// round mix slots of both) -- what calculate_round_evals records for one bivariate product claim
// (under a peer exchange the kernel reduces the RAW sums across the ranks and the host adds the values' initial contents
// once, where the one-round finalize adds them once per rank: only zero initial values mean the same thing on both paths)
bool two_round_recipe_ok(const bn::fin_fuse &f)
{
	const bn::fin_args &a = f.args;
	if (f.peer.world > 1)
		for (uint32_t v = 0; v < a.n_values; v++)
			if (!(a.init[v] == f128{0, 0})) return false;
	return a.n_terms == 2 && a.n_ret >= 1 && a.n_ret <= 2 && a.terms[0].coeff == a.terms[1].coeff;
}
bn::fin_fuse two_round_recipe(const bn::fin_fuse &fz)
{
	bn::fin_fuse f8 = fz;
	f8.args = bn::fin_args{};
	f8.args.n_terms = f8.args.n_values = f8.args.n_ret = f8.args.n_slots = 8;
	f8.args.seq = fz.args.seq;
	for (uint32_t i = 0; i < 8; i++) {
		f8.args.terms[i] = bn::fin_term{i, i, fz.args.terms[0].coeff};
		f8.args.ret_ids[i] = i;
	}
	return f8;
}
// ---- host tail ---------------------------------------------------------------------------------------------------------
// A two-round launch whose Y is at most ht_max elements per array (2^12 when the host folds on VPCLMULQDQ, else 2^8) also hands Y to the host,
// mapped into the power basis of hostmul_clmul.cpp by one more nibble-table product per element.  From then on the sumcheck
// is host arithmetic: a fold of 2 x 128 elements is 0.9 us, a round evaluation over 128 points 0.8 us (the 256-bit
// carry-less products are XORed unreduced, one reduction per sum), against ~10 us per round through the device.  The calls
// must be the expected ones (the fold of exactly these arrays, the evaluation of exactly their halves); anything else
// writes the host's folded copy back (launch_tail_writeback: the caller's memory ends up as eager execution leaves it)
// and the device path takes over again.  Not under a peer exchange (the ranks' partial sums meet on the devices there).
bool host_tail_applies(const bn_ctx *ctx, uint64_t m, uint32_t peer_world)
{
	// (under a peer exchange only for a caller that exchanges the host rounds' partial sums itself: bn_host_tail_allow_peer)
	return ctx->ht_enabled && ctx->lazy_fold && (peer_world <= 1 || ctx->ht_peer_ok) && m >= 4 && m <= ctx->ht_max;
}

struct two_round_req {
	bn::foldeval8_args fa;
	f128 z1{0, 0}, z2{0, 0};
	const void *lo[2] = {}, *hi[2] = {}; // the halves of Y, the arrays the eight sums describe
	uint64_t m = 0;                      // elements of Y
};

// Arms the launch AFTER a two-round kernel that leaves Y = (lo | hi, m elements): two folds in place, Y'' of m / 4.
void arm_two_round_next(bn_ctx *ctx, const two_round_req &rq, const bn::fin_fuse &f8, f128 *d_S)
{
	if (!ctx->arm_enabled || ctx->prof_on || ctx->tail_max_n_in || !two_round_size_ok(rq.m >> 2)) return;
	if (host_tail_applies(ctx, rq.m, f8.peer.world)) return; // (the host has the arrays: nothing more is launched for this instance)
	bn::foldeval8_args fn{};
	for (int j = 0; j < 2; j++) {
		fn.x0[j] = rq.lo[j];
		fn.x1[j] = rq.hi[j];
		fn.out[j] = const_cast<void *>(rq.lo[j]);
	}
	fn.n_in = rq.m;
	fn.n_folds = 2;
	if (host_tail_applies(ctx, rq.m >> 2, f8.peer.world)) {
		fn.mirror = (f128 *)ctx->d_tail;
		fn.phi_tab = (const uint4 *)ctx->d_phi;
		fn.tag_acc = ctx->d_ht_tag;
	}
	bn::fin_fuse fzn = f8;
	fzn.args.seq = f8.args.seq + 1;
	fzn.peer.round = f8.peer.round + 1;
	bn::arm_args aa{};
	aa.h_cmd = (const uint64_t *)&ctx->d_mail[84].lo;
	aa.h_status = (uint64_t *)&ctx->d_mail[87].lo;
	aa.d_relay = ctx->d_arm_relay;
	aa.id = ++ctx->arm_counter;
	if (bn::launch_foldeval8(ctx->stream, fn, f128{0, 0}, f128{0, 0}, d_S, &fzn, &aa) != hipSuccess) {
		(void)hipGetLastError();
		return;
	}
	bn_ctx::arm_state &am = ctx->arm;
	am.active = true;
	am.id = aa.id;
	am.n_in = fn.n_in;
	am.nf = 2;
	for (int j = 0; j < 2; j++) {
		am.x0[j] = fn.x0[j];
		am.x1[j] = fn.x1[j];
		am.out[j] = fn.out[j];
	}
	am.scale_mask = 0;
	am.seq = fzn.args.seq;
	am.peer_round = fzn.peer.world > 1 ? fzn.peer.round : 0;
	am.d_sums = d_S;
	am.recipe = recipe_bytes(fzn.args);
}

// Runs the two-round kernel for the caller's (y_1, y_inf) request `fz` (finalize recipe, mailbox sequence number and peer
// round already assigned), through an armed kernel when the one waiting on the device is exactly this launch; returns the
// caller's values in h_out and leaves the next round's quadratics in ctx->pre.
int two_round_launch(bn_ctx *ctx, const two_round_req &rq, const bn::fin_fuse &fz, const f128 *init, uint32_t n_values, const uint32_t *ret_values,
                     uint32_t n_ret, bn_f128 *h_out, f128 *d_S, std::chrono::steady_clock::time_point t_enter)
{
	hipStream_t s = ctx->stream;
	const bn::fin_fuse f8 = two_round_recipe(fz);
	volatile uint64_t *seqw = &ctx->h_mail[64].lo;
	bool got = false;
	if (ctx->arm.active) {
		bn_ctx::arm_state &am = ctx->arm;
		bool same = am.nf == rq.fa.n_folds && am.nf != 0 && am.n_in == rq.fa.n_in && am.seq == f8.args.seq && am.d_sums == d_S &&
		            am.peer_round == (f8.peer.world > 1 ? f8.peer.round : 0);
		for (int j = 0; j < 2 && same; j++) same = am.x0[j] == rq.fa.x0[j] && am.x1[j] == rq.fa.x1[j] && am.out[j] == rq.fa.out[j];
		if (same) same = recipe_bytes(f8.args) == am.recipe;
		if (!same) {
			arm_cancel(ctx);
		} else {
			const uint64_t id = am.id;
			am.active = false;
			ctx->h_mail[85].lo = rq.z1.lo;
			ctx->h_mail[85].hi = rq.z1.hi;
			ctx->h_mail[86].lo = rq.z2.lo; // (the second challenge travels in the hi_scale slot of the command block)
			ctx->h_mail[86].hi = rq.z2.hi;
			__atomic_store_n(arm_cmd(ctx), (id << 2) | 1ull, __ATOMIC_RELEASE);
			const auto t_go = std::chrono::steady_clock::now();
			arm_two_round_next(ctx, rq, f8, d_S);
			const auto t_armed = std::chrono::steady_clock::now();
			for (uint64_t spins = 0;; spins++) {
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) == f8.args.seq) {
					got = true;
					break;
				}
				if ((spins & 63) == 63) {
					const uint64_t st = __atomic_load_n(arm_status(ctx), __ATOMIC_ACQUIRE);
					if (st == (id | bn::kArmLost)) {
						arm_cancel(ctx);
						BN_HIP(hipStreamSynchronize(s));
						return bn::fail(BN_ERR_DEVICE, "device error: an armed round was only partially executed");
					}
					if (st == id) { // it gave up waiting (bounded spin) -- did it answer first?
						got = __atomic_load_n(seqw, __ATOMIC_ACQUIRE) == f8.args.seq;
						break;
					}
				}
				if (spins > (1ull << 26)) {
					// slow rather than dead (ranks that share one device under a stress mode, a profiler): let the stream drain --
					// the kernel queued behind this one leaves by its own bounded spin -- and look once more before giving up
					BN_HIP(hipStreamSynchronize(s));
					got = __atomic_load_n(seqw, __ATOMIC_ACQUIRE) == f8.args.seq;
					if (!got && __atomic_load_n(arm_status(ctx), __ATOMIC_ACQUIRE) != id) {
						arm_cancel(ctx);
						return bn::fail(BN_ERR_DEVICE, "device error: armed round kernel stopped answering");
					}
					break;
				}
			}
			if (got) {
				ctx->arm_hits++;
				ctx->arm_ns_launch += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_armed - t_go).count();
				ctx->arm_ns_parse += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_go - t_enter).count();
				ctx->arm_ns_wait += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_go).count();
			} else {
				ctx->arm_expired++;
				arm_cancel(ctx); // the one queued behind it must leave too; the round runs the ordinary way
			}
		}
	}
	if (!got) {
		{
			prof_scope ps(ctx, BN_PROF_FOLD_EVAL8);
			BN_HIP(bn::launch_foldeval8(s, rq.fa, rq.z1, rq.z2, d_S, &f8, nullptr));
		}
		if (rq.fa.n_folds) arm_two_round_next(ctx, rq, f8, d_S); // (after round 0 the first fold goes to a buffer we have not seen yet)
		uint64_t spins = 0;
		while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != f8.args.seq) {
			if (++spins > (1ull << 22)) {
				arm_cancel(ctx);
				BN_HIP(hipStreamSynchronize(s));
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != f8.args.seq)
					return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
				break;
			}
		}
	}
	if (f8.peer.world > 1 && __atomic_load_n(&ctx->h_mail[65].lo, __ATOMIC_RELAXED) == f8.peer.round) {
		arm_cancel(ctx);
		return bn::fail(BN_ERR_DEVICE, "device error: peer exchange timed out waiting for a rank");
	}
	f128 E[8];
	for (int i = 0; i < 8; i++) {
		E[i].lo = __atomic_load_n(&ctx->h_mail[i].lo, __ATOMIC_RELAXED);
		E[i].hi = __atomic_load_n(&ctx->h_mail[i].hi, __ATOMIC_RELAXED);
	}
	// slots (kernels_foldeval8.hip): 0 = a2 b2, 1 = (a0+a2)(b0+b2), 2 = a3 b3 = P(1), 3 = (a1+a3)(b1+b3) = P2, 4 = P0, 5 = Q0, 6 = Q(1), 7 = Q2
	f128 vals[bn::kFinMaxValues];
	for (uint32_t v = 0; v < n_values; v++) vals[v] = init[v];
	vals[fz.args.terms[0].value] ^= E[0] ^ E[2];
	vals[fz.args.terms[1].value] ^= E[1] ^ E[3];
	for (uint32_t r = 0; r < n_ret; r++) h_out[r] = bn_f128{vals[ret_values[r]].lo, vals[ret_values[r]].hi};
	ctx->fin_y.valid = false;
	if (rq.fa.mirror) {
		// ---- host tail: Y is in the pinned staging, in the power basis.  The kernel wrote it with plain posted stores and
		// nothing is assumed about their order against the sequence word: the tag in mailbox word 66 is the tag OF the data (a
		// mix of every element and its index, XOR the sequence number), and the staging is accepted only when what is read
		// matches it -- normally at the first look.
		bn_ctx::host_tail_state &ht = ctx->ht;
		const uint64_t *src = (const uint64_t *)ctx->h_tail;
		for (int j = 0; j < 2; j++) ht.y[j].resize(2 * rq.m);
		bool valid = false;
		for (int tries = 0; tries < 4096 && !valid; tries++) {
			if (tries == 2048) BN_HIP(hipStreamSynchronize(s)); // (a finished kernel's stores are visible: the last word on the matter)
			const uint64_t want = __atomic_load_n(&ctx->h_mail[66].lo, __ATOMIC_ACQUIRE);
			uint64_t t = f8.args.seq;
			for (int j = 0; j < 2; j++)
				for (uint64_t i = 0; i < rq.m; i++) {
					const uint64_t lo = __atomic_load_n(&src[2 * (rq.m * j + i)], __ATOMIC_RELAXED), hi = __atomic_load_n(&src[2 * (rq.m * j + i) + 1], __ATOMIC_RELAXED);
					ht.y[j][2 * i] = lo;
					ht.y[j][2 * i + 1] = hi;
					const uint64_t idx = rq.m * j + i; // (the device's mirror_mix, kernels_foldeval8.hip)
					const unsigned r1 = (unsigned)(idx & 63), r2 = (unsigned)((idx * 7 + 17) & 63);
					t ^= ((lo << r1) | (r1 ? lo >> (64 - r1) : 0)) ^ ((hi << r2) | (r2 ? hi >> (64 - r2) : 0)) ^ (idx + 1) * 0x9E3779B97F4A7C15ull;
				}
			valid = t == want;
		}
		if (!valid) return bn::fail(BN_ERR_DEVICE, "device error: the host tail's staging never became consistent");
		for (int j = 0; j < 2; j++) {
			ht.cur_lo[j] = rq.lo[j];
			ht.cur_hi[j] = rq.hi[j];
		}
		ht.cur_m = rq.m;
		ht.n_levels = 0;
		ht.evaluated = true;
		ht.active = true;
		ctx->ht_started++;
		ctx->pre.valid = false;
		ctx->pend.active = false;
		ctx->pend2.active = false;
		ctx->s_clean = true;
		ctx->two_round_launches++;
		return BN_OK;
	}
	if (rq.m == 4 && f8.peer.world <= 1) {
		// the last launch of this sumcheck: the kernel left Y's four elements per array in the mailbox (slots 32 .. 39)
		for (int j = 0; j < 2; j++) {
			ctx->fin_y.lo[j] = rq.lo[j];
			for (int q = 0; q < 4; q++) {
				ctx->fin_y.y[j][q].lo = __atomic_load_n(&ctx->h_mail[32 + 4 * j + q].lo, __ATOMIC_RELAXED);
				ctx->fin_y.y[j][q].hi = __atomic_load_n(&ctx->h_mail[32 + 4 * j + q].hi, __ATOMIC_RELAXED);
			}
		}
		ctx->fin_y.valid = true;
	}
	bn_ctx::precomp_state &pre = ctx->pre;
	pre.valid = true;
	pre.consumed = false;
	for (int j = 0; j < 2; j++) {
		pre.lo[j] = rq.lo[j];
		pre.hi[j] = rq.hi[j];
	}
	pre.m = rq.m;
	pre.P0 = E[4];
	pre.P1v = E[2];
	pre.P2 = E[3];
	pre.Q0 = E[5];
	pre.Q1v = E[6];
	pre.Q2 = E[7];
	pre.recipe = recipe_bytes(fz.args);
	ctx->pend.active = false;
	ctx->pend2.active = false;
	ctx->s_clean = true;
	ctx->two_round_launches++;
	return BN_OK;
}

// ---- weighted shadow of the MLE-check round evaluation ------------------------------------------------------------------
// The literal BivariateMLEcheckProver (v3/bivariate_mlecheck.rs:145-254, 391-520) asks, every round, for the two sums of
// a * b * eq over the halves of its arrays -- two chained products per point -- folds a and b, and folds its indicator table
// by adding the halves.  The same numbers come out of the BIVARIATE kernels (matrix-core Gram sums, fold + evaluation in one
// pass) if the backend keeps S = lambda * b (.) eq beside the caller's arrays: the indicator factorises over the variables,
// eq_r(u, x'') = eq_{r+1}(x'') * (u ? zeta : 1 - zeta), so folding S with the round challenge and multiplying its upper half
// by (1 - zeta) / zeta gives lambda (1 - zeta) * b' (.) eq' again (host/sumcheck.hpp WeightedMLEcheckProver does this above
// the trait; here it happens below it, for an unchanged caller).  All the backend needs from the table are the ratios
// rho_k = eq[2^k] / eq[0] = zeta_k / (1 - zeta_k), and the certainty that the table has that structure everywhere
// (k_check_tensor, one pass).  Whatever the caller does that is not the expected call drops the shadow; the literal
// kernels then answer as before.
constexpr uint64_t kShadowMinHalf = 1ull << 9; // (smaller instances: the three-factor kernel is as good)

// First evaluation of an instance: S = b (.) eq on both halves, in the shadow's own memory -- exactly the two element-wise
// passes the routed three-factor evaluation makes anyway, so a lone evaluation costs what it cost before.  Whether the
// table has the structure the LATER rounds rely on is only looked at when the caller's fold arrives (shadow_check_table).
int shadow_begin(bn_ctx *ctx, const void *a_lo, const void *a_hi, const void *b_lo, const void *b_hi, const void *eq, uint64_t half)
{
	bn_ctx::shadow_state &sh = ctx->shadow;
	sh.valid = false;
	sh.fold_pending = false;
	const uint32_t K = ilog2(half);
	if (((uint64_t)1 << K) != half || K > 39) return BN_OK;
	if (sh.S_cap < 2 * half) {
		if (sh.S) {
			BN_HIP(hipStreamSynchronize(ctx->stream));
			(void)hipFree(sh.S);
		}
		sh.S = nullptr;
		sh.S_cap = 0;
		if (hipMalloc(&sh.S, 2 * half * sizeof(f128)) != hipSuccess) {
			(void)hipGetLastError();
			sh.S = nullptr;
			return BN_OK; // (no shadow: the literal kernels answer)
		}
		sh.S_cap = 2 * half;
	}
	sh.a_lo = a_lo;
	sh.a_hi = a_hi;
	sh.b_lo = b_lo;
	sh.b_hi = b_hi;
	sh.half = half;
	sh.eq = eq;
	sh.eq_len = half;
	sh.eq_copy_src = nullptr;
	sh.eq_copy_dst = nullptr;
	sh.lambda = f128{1, 0};
	sh.checked = false;
	// (whether the table has the structure the later rounds rely on is looked at when the caller's first fold arrives)
	BN_HIP(bn::launch_mul9(ctx->stream, ctx->n_cu, b_lo, 1, eq, 1, 0, sh.S, half));
	BN_HIP(bn::launch_mul9(ctx->stream, ctx->n_cu, b_hi, 1, eq, 1, 0, (char *)sh.S + half * sizeof(f128), half));
	sh.valid = true;
	ctx->shadow_created++;
	return BN_OK;
}

void shadow_drop(bn_ctx *ctx)
{
	side_join(ctx); // (whatever answers from now on reads the caller's arrays on the main stream)
	if (ctx->shadow.valid) ctx->shadow_dropped++;
	ctx->shadow.valid = false;
	ctx->shadow.fold_pending = false;
}

// how a kernel-buffer slice is realised on the device
struct slice_view {
	const char *p = nullptr; // direct data
	const char *q = nullptr; // if non-null: value = p ^ q (a Local buffer defined by ADD and not materialised)
	bool zero = false;       // untouched Local buffer
	uint64_t len = 0;
};
} // namespace

} // extern "C"
namespace bnabi {
hipError_t roundeval_product_routed_pub(bn_ctx *ctx, bool scratch_free, const void *const *hi, const void *const *lo, uint32_t k, uint64_t n, f128 *d_out)
{
	return roundeval_product_routed(ctx, scratch_free, hi, lo, k, n, d_out, nullptr);
}
} // namespace bnabi
extern "C" {

int bn_kernel_launch(bn_ctx *ctx, const bn_memmap *maps, uint32_t n_maps, const bn_kop *ops, uint32_t n_ops,
                     const uint32_t *ret_values, uint32_t n_ret, uint32_t log_chunks, bn_f128 *h_out, void *d_out)
{
	BN_REQUIRE(ctx && maps && n_maps > 0, "kernel launch needs at least one mapping");
	const auto t_enter = std::chrono::steady_clock::now();
	BN_ENTER(ctx);
	uint32_t lo_c, hi_c;
	int rc = bn_log_chunks_range(maps, n_maps, &lo_c, &hi_c);
	if (rc) return rc;
	BN_REQUIRE(log_chunks == 0, "this backend records kernels with log_chunks = bn_pick_log_chunks() = 0");
	BN_REQUIRE(n_ret <= 64, "too many returned values");
	hipStream_t s = ctx->stream;

	// ---- claim groups (abi_group.cpp): k product claims over m arrays, several provers' folds waiting -- the shape piop::prove
	// issues.  Declines (handled = false) everything else, and a lone single-claim prover, which the machinery below keeps.
	if (h_out && !d_out && n_ret > 0 && ops && n_ops) {
		bool handled = false;
		rc = group_eval(ctx, maps, n_maps, ops, n_ops, ret_values, n_ret, h_out, &handled);
		if (rc || handled) return rc;
	}

	// Element-wise adds on buffers that the deferred fold neither reads nor writes commute with it: they run now and the
	// fold stays deferred (the MLE-check prover folds its indicator table -- add the upper half onto the lower -- between
	// the fold of its multilinears and the next round evaluation, v3/bivariate_mlecheck.rs:195-254).
	if (ctx->pend.active && !ctx->pend2.active && n_ret == 0 && n_ops > 0 && !ctx->tail.active) {
		bool plain = true;
		for (uint32_t o = 0; o < n_ops && plain; o++) {
			const bn_kop &op = ops[o];
			if (op.kind != BN_KOP_ADD_ASSIGN && op.kind != BN_KOP_ADD) plain = false;
			for (const bn_kslice *sl : {&op.dst, &op.src1, &op.src2}) {
				if (op.kind == BN_KOP_ADD_ASSIGN && sl == &op.src2) continue;
				if (!plain) break;
				if (sl->buf >= n_maps || maps[sl->buf].kind == BN_MAP_LOCAL || sl->off + sl->len > maps[sl->buf].len ||
				    !independent_of_pending(ctx, (const char *)maps[sl->buf].d_data + sl->off * sizeof(f128), sl->len))
					plain = false;
			}
			if (plain && maps[op.dst.buf].kind == BN_MAP_CHUNKED) plain = false; // (read-only mapping: the general path reports it)
			if (plain && (op.src1.len != op.dst.len || (op.kind == BN_KOP_ADD && op.src2.len != op.dst.len))) plain = false;
		}
		if (plain) {
			bn_ctx::shadow_state &sh = ctx->shadow;
			for (uint32_t o = 0; o < n_ops; o++) {
				const bn_kop &op = ops[o];
				char *dst = (char *)maps[op.dst.buf].d_data + op.dst.off * sizeof(f128);
				const char *s1 = (const char *)maps[op.src1.buf].d_data + op.src1.off * sizeof(f128);
				if (sh.valid) {
					// the one add the shadow expects: the table's upper half onto its lower half (in place, or onto the copy of
					// the lower half made since the last fold); any other write into the table ends the shadow
					const bool onto_table = (const void *)dst == sh.eq || ((const void *)dst == sh.eq_copy_dst && sh.eq_copy_src == sh.eq);
					BN_SHDBG("independent add: dst %p len %llu src %p | eq %p len %llu copy %p->%p", (void *)dst, (unsigned long long)op.dst.len, (const void *)s1, sh.eq,
					         (unsigned long long)sh.eq_len, sh.eq_copy_src, sh.eq_copy_dst);
					if (op.kind == BN_KOP_ADD_ASSIGN && n_ops == 1 && onto_table && 2 * op.dst.len == sh.eq_len &&
					    (const void *)s1 == (const char *)sh.eq + op.dst.len * sizeof(f128)) {
						sh.eq = dst;
						sh.eq_len = op.dst.len;
						sh.eq_copy_src = nullptr;
						sh.eq_copy_dst = nullptr;
					} else if (ranges_overlap(dst, op.dst.len, sh.eq, sh.eq_len)) {
						shadow_drop(ctx);
					}
				}
				// (with a shadow alive these adds are its caller's table folds: nothing on the main stream reads the table, so they
				// run beside the sumcheck's kernels instead of between them)
				const char *s2 = op.kind == BN_KOP_ADD ? (const char *)maps[op.src2.buf].d_data + op.src2.off * sizeof(f128) : nullptr;
				if (sh.valid) {
					ctx->side_queue.push_back({op.kind == BN_KOP_ADD ? bn_ctx::side_op::ADD : bn_ctx::side_op::ADD_ASSIGN, dst, s1, s2, op.dst.len, f128{0, 0}});
				} else if (op.kind == BN_KOP_ADD_ASSIGN) {
					BN_HIP(bn::launch_add_assign(s, dst, s1, op.dst.len));
				} else {
					BN_HIP(bn::launch_add(s, dst, s1, s2, op.dst.len));
				}
			}
			return BN_OK;
		}
	}

	// A deferred fold survives into this launch only if the kernel has the calculate_round_evals
	// shape (two bivariate-product sums, Local "lo + hi" operands, nothing written to memory); the
	// launch site below then checks that it reads exactly the folded arrays.
	// A host tail survives into this launch only if it has the same shape and wants a host result; the launch site below then
	// checks that it reads exactly the halves of the host's arrays.
	bool ht_keep = false;
	if (ctx->ht.active) {
		uint32_t n_sum = 0;
		bool pure = n_ret > 0 && h_out && !d_out && !ctx->ht.evaluated && !ctx->pend.active;
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
		ht_keep = pure && n_sum == 2;
	}
	if (!ctx->pend.active && !ht_keep) BN_FLUSH(ctx); // (deferred copies; a parked tail kernel without a fold to run; a host tail)
	if (ctx->pend.active) {
		uint32_t n_sum = 0;
		bool pure = n_ret > 0 && ctx->pend.count == 2;
		for (uint32_t o = 0; o < n_ops && pure; o++) {
			const bn_kop &op = ops[o];
			if (op.kind == BN_KOP_SUM_COMPOSITION) {
				n_sum++;
				if (!op.expr || op.expr->shape != bn_expr::PRODUCT ||
				    !(op.expr->product_vars.size() == 2 || (op.expr->product_vars.size() == 3 && ctx->shadow.valid && ctx->shadow.fold_pending)))
					pure = false;
			} else if (op.kind == BN_KOP_ADD) {
				if (op.dst.buf >= n_maps || maps[op.dst.buf].kind != BN_MAP_LOCAL) pure = false;
			} else if (op.kind != BN_KOP_DECL_VALUE) {
				pure = false;
			}
		}
		if (!pure || n_sum != 2) BN_FLUSH(ctx);
		// two folds are deferred but the launch cannot fold twice (the two-round kernel needs >= 4 elements and a host result):
		// the first one runs now, the second stays deferred for the one-round kernels
		if (ctx->pend.active && ctx->pend2.active &&
		    !(ctx->two_round && h_out && !d_out && two_round_size_ok(ctx->pend2.n))) {
			rc = flush_first_fold(ctx);
			if (rc) return rc;
		}
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
		// The materialised Locals live in the context's scratch, and so do the temporaries of a generic composition compiled into
		// passes (circuit_multipass_sum asks for local_bytes + its temporaries): the block is sized for BOTH here, once -- a later
		// growth would free the Locals under the ops that already wrote them.
		size_t temp_bytes = 0;
		for (uint32_t o = 0; o < n_ops; o++) {
			const bn_kop &op = ops[o];
			if (op.kind != BN_KOP_SUM_COMPOSITION || !op.expr || op.expr->shape == bn_expr::PRODUCT || !op.n_rows) continue;
			if (!circuit_multipass_applies(ctx, op.expr, op.rows[0].len)) continue;
			const int t = circuit_multipass_sum_temps(op.expr, false);
			if (t > 0 && (size_t)t * op.rows[0].len * sizeof(f128) > temp_bytes) temp_bytes = (size_t)t * op.rows[0].len * sizeof(f128);
		}
		char *mem = (char *)bn::ctx_scratch(ctx, local_bytes + temp_bytes);
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
	// the MLE-check shadow returns lambda times the caller's sums from the kernel: divided out here, on the host
	bool out_scaled = false;
	f128 out_mul{1, 0};
	auto scale_out = [&]() {
		if (!out_scaled || !h_out) return;
		for (uint32_t r = 0; r < n_ret; r++) {
			const f128 v = bn::mul_host(f128{h_out[r].lo, h_out[r].hi}, out_mul);
			h_out[r] = bn_f128{v.lo, v.hi};
		}
	};
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

	// A kernel with more sums than one finalize step carries (64 accumulator slots, kFinMaxTerms terms -- a prover with more than
	// sixteen claims whose evaluation the claim groups did not take: abi_group.cpp declined it or is switched off): the values so
	// far come back to the host, the slots are re-zeroed by that finalize, and the kernel goes on from those values.  Rank-local
	// (no peer exchange: the last finalize of the call reduces the totals).
	auto drain = [&]() -> int {
		bn::fin_args fa{};
		fa.n_terms = (uint32_t)terms.size();
		fa.n_values = n_values;
		fa.n_ret = n_values;
		for (size_t t = 0; t < terms.size(); t++) fa.terms[t] = terms[t];
		for (uint32_t v = 0; v < n_values; v++) {
			fa.init[v] = h_values[v];
			fa.ret_ids[v] = v;
		}
		fa.n_slots = n_slots;
		fa.seq = ++ctx->mail_seq;
		ctx->mirror.valid = false;
		BN_HIP(bn::launch_finalize(s, fa, d_S, d_rets, ctx->d_mail, nullptr));
		volatile uint64_t *seqw = &ctx->h_mail[64].lo;
		uint64_t spins = 0;
		while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != fa.seq) {
			if (++spins > (1ull << 22)) {
				BN_HIP(hipStreamSynchronize(s));
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != fa.seq) return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
				break;
			}
		}
		for (uint32_t v = 0; v < n_values; v++) {
			h_values[v].lo = __atomic_load_n(&ctx->h_mail[v].lo, __ATOMIC_RELAXED);
			h_values[v].hi = __atomic_load_n(&ctx->h_mail[v].hi, __ATOMIC_RELAXED);
		}
		terms.clear();
		n_slots = 0;
		return BN_OK;
	};
	static_assert(bn::kFinMaxValues <= bn::kFinMaxRets, "a drain returns every value");

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
			if (n_slots + 2 > 64 || terms.size() + 2 > (size_t)bn::kFinMaxTerms) {
				rc = drain();
				if (rc) return rc;
			}
			const uint32_t slot = n_slots;
			if (op.expr->shape == bn_expr::PRODUCT) {
				// fused pairing: if the NEXT sum op uses the same expression and its factors are the
				// "infinity" versions (Local = lo + hi with hi == this op's row) of this op's factors,
				// do both with one pass over the data.
				uint32_t k = (uint32_t)op.expr->product_vars.size(); // (the MLE-check shadow rewrites a three-factor request into a bivariate one)
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
								const bool peer_on = ctx->peer.active;
								if (peer_on) {
									// the finalize step of this launch reduces the returned values across the ranks (bn_peer_*)
									BN_REQUIRE(h_out && !d_out, "peer exchange: a reduced launch returns to the host");
									fz.peer.world = ctx->peer.world;
									fz.peer.rank = ctx->peer.rank;
									for (uint32_t w = 0; w < ctx->peer.world; w++) fz.peer.box[w] = (uint64_t *)ctx->peer.box[w];
									fz.peer.round = ++ctx->peer.round;
									fz.peer.stress = ctx->peer.stress;
								}
								hipError_t fe = hipErrorNotSupported;
								// ---- a host tail can also START here, without a kernel: the arrays of this evaluation lie in the context's own pinned
								// scratch (bn_host_scratch: host memory the device reads through a mapping -- the residual instance of the sharded
								// prover, a handful of elements the ranks have just exchanged on the host), the library reads them where they are
								if (!ctx->ht.active && host_tail_applies(ctx, 2 * row_len, peer_on ? ctx->peer.world : 1) && k == 2 && !peer_on && h_out && !d_out && !ctx->pend.active &&
								    lo[0] && lo[1] && lo[0] != lo[1] && !out_scaled && two_round_recipe_ok(fz) && ctx->pend_copies.empty()) {
									const char *d0 = (const char *)(ctx->d_mail + 96), *d1 = d0 + 32 * sizeof(f128);
									auto inside = [&](const void *p) { return (const char *)p >= d0 && (const char *)p + row_len * sizeof(f128) <= d1; };
									if (inside(lo[0]) && inside(hi[0]) && inside(lo[1]) && inside(hi[1]) && hipStreamSynchronize(s) == hipSuccess) { // (nothing enqueued is still writing there: microseconds, against three device round trips)
										bn_ctx::host_tail_state &ht = ctx->ht;
										auto host_view = [&](const void *p) { return (const f128 *)((const char *)(ctx->h_mail + 96) + ((const char *)p - d0)); };
										for (int j = 0; j < 2; j++) {
											ht.y[j].resize(4 * row_len);
											for (uint64_t i = 0; i < row_len; i++) {
												const bn::hp128 a = bn::hostpoly_from_tower(host_view(lo[j])[i]), b = bn::hostpoly_from_tower(host_view(hi[j])[i]);
												ht.y[j][2 * i] = a.lo;
												ht.y[j][2 * i + 1] = a.hi;
												ht.y[j][2 * (row_len + i)] = b.lo;
												ht.y[j][2 * (row_len + i) + 1] = b.hi;
											}
											ht.cur_lo[j] = lo[j];
											ht.cur_hi[j] = hi[j];
										}
										ht.cur_m = 2 * row_len;
										ht.n_levels = 0;
										ht.evaluated = false;
										ht.active = true;
										ctx->ht_started++;
									}
								}
								// ---- host tail: the halves of exactly the arrays the host holds -- answered here, no launch
								if (ctx->ht.active) {
									bn_ctx::host_tail_state &ht = ctx->ht;
									auto is = [&](int f, int j) { return lo[f] == ht.cur_lo[j] && hi[f] == ht.cur_hi[j]; };
									// (a caller that exchanges the host rounds' partial sums itself -- bn_host_tail_allow_peer -- XORs the ranks' RETURNED
									// values: a non-zero initial value would be counted once per rank, so only zero initial values are answered here)
									bool init_ok = true;
									if (ctx->ht_peer_ok)
										for (uint32_t v = 0; v < n_values; v++) init_ok = init_ok && h_values[v] == f128{0, 0};
									const bool match = k == 2 && !ht.evaluated && !peer_on && h_out && !d_out && 2 * row_len == ht.cur_m && lo[0] && lo[1] &&
									                   ((is(0, 0) && is(1, 1)) || (is(0, 1) && is(1, 0))) && two_round_recipe_ok(fz) && init_ok;
									if (match) {
										bn::hp128 p1, pi;
										bn::hostpoly_round_sums(reinterpret_cast<const bn::hp128 *>(ht.y[0].data()), reinterpret_cast<const bn::hp128 *>(ht.y[1].data()), row_len, &p1, &pi);
										f128 s1 = bn::hostpoly_to_tower(p1), si = bn::hostpoly_to_tower(pi);
										const f128 cf = fz.args.terms[0].coeff;
										if (!(cf == f128{1, 0})) {
											s1 = bn::mul_host(cf, s1);
											si = bn::mul_host(cf, si);
										}
										f128 vals[bn::kFinMaxValues];
										for (uint32_t v = 0; v < n_values; v++) vals[v] = h_values[v];
										vals[fz.args.terms[0].value] ^= s1;
										vals[fz.args.terms[1].value] ^= si;
										for (uint32_t r = 0; r < n_ret; r++) h_out[r] = bn_f128{vals[ret_values[r]].lo, vals[ret_values[r]].hi};
										ht.evaluated = true;
										--ctx->mail_seq; // (no launch, no mailbox traffic)
										ctx->s_clean = was_clean_or_zeroed;
										ctx->ht_rounds++;
										return BN_OK;
									}
									BN_FLUSH(ctx); // not the expected evaluation: the device catches up first
								}
								// ---- MLE-check shape a * b * eq (one factor the same at both points): the weighted shadow
								if (k == 3 && ctx->shadow_enabled && h_out && !d_out && !peer_on && ctx->lazy_fold) {
									int same = -1, n_same = 0;
									for (int j = 0; j < 3; j++)
										if (!lo[j]) {
											same = j;
											n_same++;
										}
									bn_ctx::shadow_state &sh = ctx->shadow;
									if (n_same == 1) {
										int x = (same + 1) % 3, y = (same + 2) % 3;
										auto is = [&](int xa, int yb) { return hi[xa] == sh.a_hi && lo[xa] == sh.a_lo && hi[yb] == sh.b_hi && lo[yb] == sh.b_lo; };
										bool use = sh.valid && row_len == sh.half && sh.eq_len == row_len && hi[same] == sh.eq && (is(x, y) || is(y, x));
										if (use && !is(x, y)) std::swap(x, y);
										if (use && sh.fold_pending != ctx->pend.active) use = false; // (the deferred fold must be the one the shadow folds with)
										bool zero_init = true; // (lambda is divided out of the RETURNED values: they must be pure sums)
										for (uint32_t v = 0; v < n_values; v++) zero_init = zero_init && h_values[v] == f128{0, 0};
										if (!zero_init) use = false;
										BN_SHDBG("eval: rows=%llu valid=%d half=%llu eq_len=%llu eq %p/%p fp=%d pend=%d use=%d", (unsigned long long)row_len, (int)sh.valid,
										         (unsigned long long)sh.half, (unsigned long long)sh.eq_len, hi[same], sh.eq, (int)sh.fold_pending, (int)ctx->pend.active, (int)use);
										if (!use) {
											if (ctx->pend.active) BN_FLUSH(ctx); // a fold the shadow is not party to runs first (and ends the shadow)
											shadow_drop(ctx);
											if (zero_init && row_len >= kShadowMinHalf && row_len >= sh.blocked_below && hi[x] != hi[y]) {
												sh.blocked_below = 0;
												rc = shadow_begin(ctx, lo[x], hi[x], lo[y], hi[y], hi[same], row_len);
												if (rc) return rc;
												use = sh.valid;
											}
										}
										if (use) {
											// The request becomes the bivariate round of (a, S) and takes the ordinary path below (matrix-core /
											// 9-lane kernels, fold + evaluation in one pass, armed small rounds); lambda is divided out of the two
											// returned values on the host, so that the finalize recipe is the caller's own in every round.
											out_scaled = true;
											out_mul = bn::invert_tower(sh.fold_pending ? sh.lambda_next : sh.lambda);
											if (sh.fold_pending) {
												// the caller's deferred fold of (a, b): b folds now, in a launch of its own on the side stream (nothing of
												// the sumcheck reads it again before the caller does); the deferred fold becomes that of (a, S), with the
												// upper half of S levelled by (1 - zeta) / zeta
												bn_ctx::pending_fold &pf = ctx->pend;
												// (queued: launched once this round's kernel has its challenge, while the host would only spin)
												ctx->side_queue.push_back({bn_ctx::side_op::FOLD, pf.x0[sh.ib], pf.x1[sh.ib],
												                           pf.src0[sh.ib] != pf.x0[sh.ib] ? pf.src0[sh.ib] : nullptr, pf.n, pf.z});
												bn_ctx::pending_fold q = pf;
												q.x0[0] = pf.x0[sh.ia];
												q.x1[0] = pf.x1[sh.ia];
												q.src0[0] = pf.src0[sh.ia];
												q.x0[1] = sh.S;
												q.x1[1] = (const char *)sh.S + pf.n * sizeof(f128);
												q.src0[1] = sh.S;
												q.scale_mask = 2;
												q.hi_scale = sh.hi_scale;
												pf = q;
												sh.fold_pending = false;
												sh.lambda = sh.lambda_next;
											}
											const void *ah = hi[x], *al = lo[x];
											hi[0] = ah;
											lo[0] = al;
											hi[1] = (const char *)sh.S + row_len * sizeof(f128);
											lo[1] = sh.S;
											hi[2] = lo[2] = nullptr;
											k = 2;
											ctx->shadow_rounds++;
										}
									}
								}
								if (ctx->pend.active) {
									// fold + evaluate in one pass: this launch reads the halves of exactly the two
									// arrays the deferred fold writes (evals_1 directly behind evals_0, in place)
									const bool two = ctx->pend2.active;
									const bn_ctx::pending_fold &pf = two ? ctx->pend2 : ctx->pend; // the LAST deferred fold: its output is what this launch reads
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
											if ((pf.scale_mask >> i) & 1) fa.scale_mask |= 1u << j;
										}
										fa.hi_scale = pf.hi_scale;
										const uint64_t n_in = 2 * pf.n;
										const bool two_ok = ctx->two_round && h_out && !d_out && !fa.scale_mask && !ctx->tail_max_n_in && !ctx->tail.active &&
										                    two_round_recipe_ok(fz) && two_round_size_ok(pf.n);
										// (h) the next-round quadratics of the previous two-round launch describe exactly the arrays this fold
										// folds: the host answers -- y(z) = c0 + z c1 + z^2 c2 at the fold's challenge -- and the fold stays deferred
										if (!two && h_out && !d_out && pre_matches(ctx->pre, pf) && two_round_recipe_ok(fz) && recipe_bytes(fz.args) == ctx->pre.recipe) {
											const bn_ctx::precomp_state &pre = ctx->pre;
											const f128 z = pf.z, zz = bn::mul_host(z, z);
											const f128 y1 = pre.P0 ^ bn::mul_host(z, pre.P0 ^ pre.P1v ^ pre.P2) ^ bn::mul_host(zz, pre.P2);
											const f128 yi = pre.Q0 ^ bn::mul_host(z, pre.Q0 ^ pre.Q1v ^ pre.Q2) ^ bn::mul_host(zz, pre.Q2);
											f128 vals[bn::kFinMaxValues];
											for (uint32_t v = 0; v < n_values; v++) vals[v] = h_values[v];
											vals[fz.args.terms[0].value] ^= y1;
											vals[fz.args.terms[1].value] ^= yi;
											for (uint32_t r = 0; r < n_ret; r++) h_out[r] = bn_f128{vals[ret_values[r]].lo, vals[ret_values[r]].hi};
											ctx->pre.consumed = true;
											--ctx->mail_seq; // (no launch, no mailbox traffic: the armed kernel behind us keeps its sequence number)
											if (peer_on) --ctx->peer.round;
											ctx->s_clean = was_clean_or_zeroed;
											ctx->two_round_hosted++;
											return BN_OK;
										}
										// (t) two rounds per launch (kernels_foldeval8.hip): both deferred folds, or the one, then the eight sums
										if (two_ok) {
											const bn_ctx::pending_fold &p1 = ctx->pend;
											two_round_req rq{};
											for (uint32_t j = 0; j < 2; j++) {
												const uint32_t i = perm ? 1 - j : j;
												rq.fa.x0[j] = p1.src0[i];
												rq.fa.x1[j] = p1.x1[i];
												rq.fa.out[j] = p1.x0[i];
												rq.lo[j] = pf.x0[i];
												rq.hi[j] = (const char *)pf.x0[i] + (pf.n / 2) * sizeof(f128);
											}
											rq.fa.n_in = 2 * p1.n;
											rq.fa.n_folds = two ? 2 : 1;
											rq.z1 = p1.z;
											rq.z2 = two ? pf.z : f128{0, 0};
											rq.m = pf.n;
											if (host_tail_applies(ctx, rq.m, fz.peer.world)) {
												rq.fa.mirror = (f128 *)ctx->d_tail;
												rq.fa.phi_tab = (const uint4 *)ctx->d_phi;
													rq.fa.tag_acc = ctx->d_ht_tag;
											}
											return two_round_launch(ctx, rq, fz, h_values.data(), n_values, ret_values, n_ret, h_out, d_S + slot, t_enter);
										}
										if (two) { // (two_round_size_ok was checked at entry; something else rules the two-round kernel out)
											rc = flush_first_fold(ctx);
											if (rc) return rc;
										}
										if (fa.scale_mask && ctx->tail.active) { // the resident tail kernel folds without a scale
											rc = tail_cancel(ctx);
											if (rc) return rc;
										}
										// arm the kernel of the round after (fa_, n_in_, seq_) behind whatever runs that round
										// (arm.hpp): same arrays in place, half the size, same recipe, next sequence number
										auto arm_next = [&](const bn::foldeval_args &fa_, uint64_t n_in_, const bn::fin_fuse &fz_) {
											const uint64_t n_next = n_in_ >> 1;
											// only latency-shaped rounds: a launch is nothing next to a kernel of 2^20 elements, and the
											// host waits for long kernels with a stream synchronisation, which an armed kernel would hold up
											const bool mfma_next = bn::mfma_applies(ctx->n_cu, n_next >> 2);
											if (ctx->arm_enabled && !ctx->prof_on && h_out && !d_out && ctx->two_round && !fa_.scale_mask && !ctx->tail_max_n_in &&
											    two_round_recipe_ok(fz_) && two_round_size_ok(n_next >> 1)) {
												// the next round will be a two-round launch with ONE fold (its arrays: the halves written now)
												bn::foldeval8_args f8a{};
												for (uint32_t j = 0; j < 2; j++) {
													f8a.x0[j] = fa_.out[j];
													f8a.x1[j] = (const char *)fa_.out[j] + (n_next >> 1) * sizeof(f128);
													f8a.out[j] = fa_.out[j];
												}
												f8a.n_in = n_next;
												f8a.n_folds = 1;
												if (host_tail_applies(ctx, n_next >> 1, fz_.peer.world)) {
													f8a.mirror = (f128 *)ctx->d_tail;
													f8a.phi_tab = (const uint4 *)ctx->d_phi;
														f8a.tag_acc = ctx->d_ht_tag;
												}
												bn::fin_fuse fzn = two_round_recipe(fz_);
												fzn.args.seq = fz_.args.seq + 1;
												fzn.peer.round = fz_.peer.round + 1;
												bn::arm_args aa{};
												aa.h_cmd = (const uint64_t *)&ctx->d_mail[84].lo;
												aa.h_status = (uint64_t *)&ctx->d_mail[87].lo;
												aa.d_relay = ctx->d_arm_relay;
												aa.id = ++ctx->arm_counter;
												if (bn::launch_foldeval8(s, f8a, f128{0, 0}, f128{0, 0}, d_S + slot, &fzn, &aa) != hipSuccess) {
													(void)hipGetLastError();
													return;
												}
												bn_ctx::arm_state &am = ctx->arm;
												am.active = true;
												am.id = aa.id;
												am.n_in = n_next;
												am.nf = 1;
												for (uint32_t j = 0; j < 2; j++) {
													am.x0[j] = f8a.x0[j];
													am.x1[j] = f8a.x1[j];
													am.out[j] = f8a.out[j];
												}
												am.scale_mask = 0;
												am.seq = fzn.args.seq;
												am.peer_round = fzn.peer.world > 1 ? fzn.peer.round : 0;
												am.d_sums = d_S + slot;
												am.recipe = recipe_bytes(fzn.args);
												return;
											}
											if (!ctx->arm_enabled || ctx->prof_on || !h_out || d_out || n_next < 4 || (n_next & 3) ||
											    n_next > (mfma_next ? arm_max_in_mfma() : kArmMaxIn) || (mfma_next && fa_.scale_mask == 3) || ctx->tail_max_n_in)
												return;
											bn_ctx::arm_state &am = ctx->arm;
											bn::foldeval_args fn{};
											for (uint32_t j = 0; j < 2; j++) {
												fn.x0[j] = fa_.out[j];
												fn.x1[j] = (const char *)fa_.out[j] + (n_next >> 1) * sizeof(f128);
												fn.out[j] = fa_.out[j];
											}
											fn.scale_mask = fa_.scale_mask;
											bn::fin_fuse fzn = fz_;
											fzn.args.seq = fz_.args.seq + 1;
											fzn.peer.round = fz_.peer.round + 1; // (unused unless fz_.peer.world > 1)
											bn::arm_args aa{};
											aa.h_cmd = (const uint64_t *)&ctx->d_mail[84].lo;
											aa.h_status = (uint64_t *)&ctx->d_mail[87].lo;
											aa.d_relay = ctx->d_arm_relay;
											aa.id = ++ctx->arm_counter;
											const hipError_t ae = mfma_next ? bn::launch_foldeval_mfma(s, ctx->n_cu, fn, n_next, f128{0, 0}, d_S + slot, &fzn, &aa)
											                                : bn::launch_foldeval9(s, ctx->n_cu, fn, n_next, f128{0, 0}, d_S + slot, &fzn, &aa);
											if (ae != hipSuccess) {
												(void)hipGetLastError();
												return;
											}
											am.active = true;
											am.id = aa.id;
											am.n_in = n_next;
											am.nf = 0;
											for (uint32_t j = 0; j < 2; j++) {
												am.x0[j] = fn.x0[j];
												am.x1[j] = fn.x1[j];
												am.out[j] = fn.out[j];
											}
											am.scale_mask = fn.scale_mask;
											am.seq = fzn.args.seq;
											am.peer_round = fzn.peer.world > 1 ? fzn.peer.round : 0;
											am.d_sums = d_S + slot;
											am.recipe = recipe_bytes(fzn.args);
										};
										// (a0) the kernel of this round is already on the device, armed: hand it z
										if (ctx->arm.active) {
											bn_ctx::arm_state &am = ctx->arm;
											bool same = am.nf == 0 && h_out && !d_out && n_in == am.n_in && fa.scale_mask == am.scale_mask && fz.args.seq == am.seq &&
											            d_S + slot == am.d_sums && am.peer_round == (fz.peer.world > 1 ? fz.peer.round : 0);
											for (uint32_t j = 0; j < 2 && same; j++)
												same = fa.x0[j] == am.x0[j] && fa.x1[j] == am.x1[j] && fa.out[j] == am.out[j];
											if (same) same = recipe_bytes(fz.args) == am.recipe;
											if (!same) {
												arm_cancel(ctx);
											} else {
												const uint64_t id = am.id;
												am.active = false;
												ctx->h_mail[85].lo = pf.z.lo;
												ctx->h_mail[85].hi = pf.z.hi;
												ctx->h_mail[86].lo = pf.hi_scale.lo;
												ctx->h_mail[86].hi = pf.hi_scale.hi;
												__atomic_store_n(arm_cmd(ctx), (id << 2) | 1ull, __ATOMIC_RELEASE);
												const auto t_go = std::chrono::steady_clock::now();
												arm_next(fa, n_in, fz); // the round after this one queues up while this one runs
												const auto t_armed = std::chrono::steady_clock::now();
												rc = side_run_queue(ctx); // (and so does the side work of a shadowed MLE-check)
												if (rc) return rc;
												volatile uint64_t *seqw = &ctx->h_mail[64].lo;
												bool got = false;
												for (uint64_t spins = 0;; spins++) {
													if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) == fz.args.seq) { got = true; break; }
													if ((spins & 63) == 63) {
														const uint64_t st = __atomic_load_n(arm_status(ctx), __ATOMIC_ACQUIRE);
														if (st == (id | bn::kArmLost)) { // a workgroup left a round that went ahead without it
															arm_cancel(ctx);
															BN_HIP(hipStreamSynchronize(s));
															return bn::fail(BN_ERR_DEVICE, "device error: an armed round was only partially executed");
														}
														if (st == id) {
															// the kernel gave up waiting (bounded spin) -- did it answer first?
															got = __atomic_load_n(seqw, __ATOMIC_ACQUIRE) == fz.args.seq;
															break;
														}
													}
													if (spins > (1ull << 26)) {
														// slow rather than dead (ranks that share one device under a stress mode, a profiler): let the
														// stream drain -- the kernel queued behind this one leaves by its own bounded spin -- and look
														// once more before giving up
														BN_HIP(hipStreamSynchronize(s));
														got = __atomic_load_n(seqw, __ATOMIC_ACQUIRE) == fz.args.seq;
														if (!got && __atomic_load_n(arm_status(ctx), __ATOMIC_ACQUIRE) != id) {
															arm_cancel(ctx);
															return bn::fail(BN_ERR_DEVICE, "device error: armed round kernel stopped answering");
														}
														break;
													}
												}
												if (got) {
													if (peer_on && __atomic_load_n(&ctx->h_mail[65].lo, __ATOMIC_RELAXED) == fz.peer.round) {
														arm_cancel(ctx);
														return bn::fail(BN_ERR_DEVICE, "device error: peer exchange timed out waiting for a rank");
													}
													for (uint32_t r = 0; r < n_ret; r++) {
														h_out[r].lo = __atomic_load_n(&ctx->h_mail[r].lo, __ATOMIC_RELAXED);
														h_out[r].hi = __atomic_load_n(&ctx->h_mail[r].hi, __ATOMIC_RELAXED);
													}
													ctx->pend.active = false;
													ctx->s_clean = true;
													ctx->arm_hits++;
													ctx->arm_ns_launch += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_armed - t_go).count();
													ctx->arm_ns_parse += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_go - t_enter).count();
													ctx->arm_ns_wait += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_go).count();
													scale_out();
													return BN_OK;
												}
												// it left without running the round: the one queued behind it must leave too, then
												// this round runs the normal way
												ctx->arm_expired++;
												arm_cancel(ctx);
											}
										}
										// (a) a resident tail kernel is parked for exactly this round: hand it z
										if (ctx->tail.active) {
											bn_ctx::tail_state &tl = ctx->tail;
											const bool same = !peer_on && h_out && !d_out && n_in == tl.n_in_next && fa.x0[0] == fa.out[0] && fa.x0[1] == fa.out[1] &&
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
													scale_out();
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
										if (fe == hipErrorNotSupported && !peer_on && !fa.scale_mask && h_out && !d_out && ctx->tail_max_n_in && n_in <= ctx->tail_max_n_in && n_in >= 8) {
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
											if (fe == hipSuccess) {
												ctx->pend.active = false;
												arm_next(fa, n_in, fz);
											}
										}
									} else if (ctx->tail.active) {
										rc = tail_cancel(ctx);
										if (rc) return rc;
									}
									if (ctx->pend.active) BN_FLUSH(ctx);
								}
								// round 0 of a small sumcheck, nothing to fold: the two-round kernel on the inputs themselves -- when the
								// number of variables is even, so that the chain of two-round launches ends on four elements (with an
								// odd number this round runs alone and the chain starts with the first fold)
								if (fe == hipErrorNotSupported && !out_scaled && !ctx->pend.active && k == 2 && lo[0] && lo[1] && lo[0] != lo[1] && ctx->two_round && ctx->lazy_fold && h_out && !d_out &&
								    !ctx->tail_max_n_in && two_round_recipe_ok(fz) && two_round_size_ok(2 * row_len) && (ilog2(2 * row_len) & 1) == 0) {
									two_round_req rq{};
									for (uint32_t j = 0; j < 2; j++) {
										rq.fa.x0[j] = lo[j];
										rq.fa.x1[j] = hi[j];
										rq.fa.out[j] = nullptr;
										rq.lo[j] = lo[j];
										rq.hi[j] = hi[j];
									}
									rq.fa.n_in = 2 * row_len;
									rq.fa.n_folds = 0;
									rq.m = 2 * row_len;
									if (host_tail_applies(ctx, rq.m, fz.peer.world)) {
										rq.fa.mirror = (f128 *)ctx->d_tail;
										rq.fa.phi_tab = (const uint4 *)ctx->d_phi;
													rq.fa.tag_acc = ctx->d_ht_tag;
									}
									return two_round_launch(ctx, rq, fz, h_values.data(), n_values, ret_values, n_ret, h_out, d_S + slot, t_enter);
								}
								if (fe == hipErrorNotSupported) {
									prof_scope ps(ctx, k == 2 && bn::mfma_applies(ctx->n_cu, row_len) ? BN_PROF_ROUND_EVAL_MFMA : BN_PROF_ROUND_EVAL);
									fe = roundeval_product_routed(ctx, !need_materialise, hi, lo, k, row_len, d_S + slot, &fz);
								}
								if (fe == hipSuccess) {
									in_kernel = true;
									finalized_in_kernel = true;
									fused_seq = fz.args.seq;
								} else if (fe != hipErrorNotSupported) {
									return bn::hip_fail(fe, "launch_roundeval_product (fused finalize)");
								} else {
									if (h_out) --ctx->mail_seq;
									if (peer_on) --ctx->peer.round;
								}
							}
							if (!in_kernel) {
								BN_FLUSH(ctx);
								prof_scope ps(ctx, BN_PROF_ROUND_EVAL);
								BN_HIP(roundeval_product_routed(ctx, !need_materialise, hi, lo, k, row_len, d_S + slot, nullptr));
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
				if (circuit_multipass_applies(ctx, op.expr, row_len)) {
					// compiled into passes of the throughput kernels (abi_circuit.cpp): the sums arrive in both slots of the pair
					BN_FLUSH(ctx);
					rc = circuit_multipass_sum(ctx, op.expr, rows.data(), row_len, nullptr, d_S + slot, need_materialise ? local_bytes : 0);
					if (rc != kCircuitDeclined) {
						if (rc) return rc;
						terms.push_back(bn::fin_term{op.value, slot, f128{op.scalar.lo, op.scalar.hi}});
						terms.push_back(bn::fin_term{op.value, slot + 1, f128{op.scalar.lo, op.scalar.hi}});
						n_slots += 2;
						break;
					}
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

	rc = flush_pending(ctx, /*keep_tail=*/true, /*publish_tiny=*/false, /*keep_shadow=*/true); // (a launch that ended up reading nothing)
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
	bool peer_standalone = false;
	if (!finalized_in_kernel) {
		bn::fin_peer pr{};
		if (ctx->peer.active) {
			// (several batched compositions, or any other shape the fused finalize does not cover: the stand-alone finalize
			// kernel does the same reduction)
			BN_REQUIRE(h_out && !d_out, "peer exchange: a reduced launch returns to the host");
			pr.world = ctx->peer.world;
			pr.rank = ctx->peer.rank;
			for (uint32_t w = 0; w < ctx->peer.world; w++) pr.box[w] = (uint64_t *)ctx->peer.box[w];
			pr.round = ++ctx->peer.round;
			pr.stress = ctx->peer.stress;
			peer_standalone = true;
		}
		BN_HIP(bn::launch_finalize(s, fa, d_S, rets, ctx->d_mail, &pr));
	}
	ctx->s_clean = true; // stream-ordered: the next launch on this stream sees zeroed slots
	rc = side_run_queue(ctx); // (side work of a shadowed MLE-check: launched while the host would only wait for the result)
	if (rc) return rc;
	if (h_out) {
		// spin on the sequence word the kernel publishes after the values (fine-grained host memory)
		volatile uint64_t *seqw = &ctx->h_mail[64].lo;
		const uint64_t want = fa.seq;
		uint64_t spins = 0;
		while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != want) {
			if (++spins > (1ull << 22)) {
				// not there yet: fall back to a stream sync so device errors surface instead of hanging
				// (an armed kernel queued behind this launch would sit out its whole timeout inside that sync)
				arm_cancel(ctx);
				BN_HIP(hipStreamSynchronize(s));
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != want)
					return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
				break;
			}
		}
		if (ctx->peer.active && (finalized_in_kernel || peer_standalone) && __atomic_load_n(&ctx->h_mail[65].lo, __ATOMIC_RELAXED) == ctx->peer.round) {
			arm_cancel(ctx);
			return bn::fail(BN_ERR_DEVICE, "device error: peer exchange timed out waiting for a rank");
		}
		for (uint32_t r = 0; r < n_ret; r++) {
			h_out[r].lo = __atomic_load_n(&ctx->h_mail[r].lo, __ATOMIC_RELAXED);
			h_out[r].hi = __atomic_load_n(&ctx->h_mail[r].hi, __ATOMIC_RELAXED);
		}
		scale_out();
	}
	return BN_OK;
}

// A small region of fine-grained pinned host memory that the device can read directly (32
// elements): inputs of a few elements can be handed to kernels without an upload.  Not part of the
// reference interface (used for the residual instance of the sharded prover).
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

} // extern "C"
