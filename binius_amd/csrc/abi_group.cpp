// binius_amd/csrc/abi_group.cpp -- claim groups: the deferral and dispatch of the call shape the reference's PCS prover issues.
//
// piop::prove builds ONE BivariateSumcheckProver per size out of all committed multilinears of that size and their
// ring-switch transparents (core/src/piop/prove.rs:271-287): k product claims over m multilinears, a multilinear possibly in
// several claims or in none; the provers run front-loaded in one batch (protocols/sumcheck/prove/front_loaded.rs:122-155):
//
//     execute(P_1) .. execute(P_p)  |  challenge  |  fold(P_1) .. fold(P_p)  |  (FRI: nothing, or fri_fold + commit)  |  execute(P_1) ..
//
// and each execute records  SUM(c_0, at 1) .. SUM(c_{k-1}, at 1), ADD x m, SUM(c_0, at inf) .. SUM(c_{k-1}, at inf)
// (protocols/sumcheck/v3/bivariate_product.rs:345-399).  The single-claim machinery of abi.cpp / abi_kernels.cpp (ONE deferred
// fold per context, two arrays, one claim) drops every one of these to eager launches.  Here
//
//   * a fold batch is deferred PER PROVER: any number of batches wait side by side as long as their arrays are disjoint
//     (bn_ctx::group_state::folds);
//   * a round evaluation of that shape is parsed into arrays (lo, hi) and claims (a, b); a matching of the claim graph becomes
//     fold + evaluate jobs (kind 0: each folds its two arrays), the arrays left over fold-only jobs (kind 3), every other claim
//     an evaluate job (kind 1) -- chained behind the jobs that fold what it reads, inside the same launch (plan_chained);
//   * ONE launch (kernels_group.hip) carries the jobs of the calling prover AND of every other prover whose deferred fold
//     is waiting and whose claims are known from its last evaluation (a session); the others' raw sums are kept and answer
//     their execute() without a launch;
//   * the batch coefficients are applied on the host (value = init + sum_c coeff_c * S_c, compute/src/cpu/layer.rs:512).
//
// One invalidation rule: a call that reads or writes device memory the deferred folds touch, or that the library cannot see
// through, runs them first as plain fold launches (group_flush / group_flush_touching) -- the caller's memory is then exactly
// what eager execution leaves -- and forgets the sums computed ahead.  A session is only a description (pointers and claim
// indices); a stale one costs a wasted evaluation, never a wrong answer: sums computed ahead are used only for a request that
// names exactly the arrays they were computed from, and die with any write into those arrays.
#include <algorithm>
#include <unordered_map>

#include <chrono>

#include "abi_common.hpp"
#include "hostmul.hpp"

namespace bnabi {

namespace {
// BN_GROUP_PROF=1: laps of the host's clock, charged to the phases of bn_ctx::group_state
struct phase_clock {
	bn_ctx::group_state &g;
	std::chrono::steady_clock::time_point t;
	explicit phase_clock(bn_ctx::group_state &gs) : g(gs)
	{
		if (g.prof) t = std::chrono::steady_clock::now();
	}
	void lap(int phase)
	{
		if (!g.prof) return;
		const auto now = std::chrono::steady_clock::now();
		g.prof_ns[phase] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t).count();
		g.prof_calls[phase]++;
		t = now;
	}
};
constexpr uint32_t kMaxArrays = (uint32_t)bn::kGroupMaxArrays; // multilinears of a prover (keccak's PCS prover has >= 100 of one size)
constexpr uint32_t kMaxClaims = (uint32_t)bn::kGroupMaxClaims;
constexpr size_t kMaxSessions = 16;

inline const char *at(const void *p, uint64_t elems) { return (const char *)p + elems * sizeof(f128); }

bool fold_touches(const bn_ctx::group_fold &f, const void *p, uint64_t n)
{
	if (!f.hull_hits(p, n)) return false;
	for (uint32_t i = 0; i < f.count; i++)
		if (ranges_overlap(p, n, f.x0[i], f.n) || ranges_overlap(p, n, f.x1[i], f.n) || ranges_overlap(p, n, f.src0[i], f.n)) return true;
	return false;
}

// a deferred fold as plain launches (kFoldBatchMax arrays each: the batch rides in the kernel arguments)
int launch_fold(bn_ctx *ctx, const bn_ctx::group_fold &f)
{
	uint32_t at = 0;
	// (more than one launch's worth: 128 arrays at a time through the wide form)
	while (f.count - at > (uint32_t)bn::kFoldBatchMax) {
		const uint32_t c = std::min<uint32_t>(f.count - at, (uint32_t)bn::kFoldWideMax);
		static bn::fold_batch_wide fw_zero{};
		bn::fold_batch_wide fw = fw_zero;
		for (uint32_t i = 0; i < c; i++) {
			fw.x0[i] = f.x0[at + i];
			fw.x1[i] = f.x1[at + i];
			fw.src0[i] = f.src0[at + i] != f.x0[at + i] ? f.src0[at + i] : nullptr;
		}
		prof_scope ps(ctx, BN_PROF_FOLD);
		BN_HIP(bn::launch_extrapolate_line_wide(ctx->stream, ctx->n_cu, fw, c, f.n, f.z));
		at += c;
	}
	for (; at < f.count; at += (uint32_t)bn::kFoldBatchMax) {
		const uint32_t c = std::min<uint32_t>(f.count - at, (uint32_t)bn::kFoldBatchMax);
		bn::fold_batch fb{};
		for (uint32_t i = 0; i < c; i++) {
			fb.x0[i] = f.x0[at + i];
			fb.x1[i] = f.x1[at + i];
			fb.src0[i] = f.src0[at + i] != f.x0[at + i] ? f.src0[at + i] : nullptr; // (absorbed copy: read there, write here)
		}
		prof_scope ps(ctx, BN_PROF_FOLD);
		BN_HIP(bn::launch_extrapolate_line_batch(ctx->stream, ctx->n_cu, fb, c, f.n, f.z));
	}
	ctx->grp.flushed_folds++;
	return BN_OK;
}

// the pinned tables, accumulator slots and value mailbox of the group launches (allocated with the first of them)
int res_alloc(bn_ctx *ctx)
{
	auto &g = ctx->grp;
	if (g.h_tables && g.d_S && g.h_gmail) return BN_OK;
	if (!g.h_tables) {
		if (hipHostMalloc(&g.h_tables, sizeof(bn::group_tables), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
			(void)hipGetLastError();
			g.h_tables = nullptr;
			return bn::fail(BN_ERR_ALLOC, "claim groups: pinned job table");
		}
		std::memset(g.h_tables, 0, sizeof(bn::group_tables));
		BN_HIP(hipHostGetDevicePointer(&g.d_tables, g.h_tables, 0));
	}
	if (!g.h_gmail) {
		if (hipHostMalloc((void **)&g.h_gmail, sizeof(f128) * bn::kGroupMaxSlots, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
			(void)hipGetLastError();
			g.h_gmail = nullptr;
			return bn::fail(BN_ERR_ALLOC, "claim groups: pinned value mailbox");
		}
		std::memset(g.h_gmail, 0, sizeof(f128) * bn::kGroupMaxSlots);
		BN_HIP(hipHostGetDevicePointer((void **)&g.d_gmail, g.h_gmail, 0));
	}
	if (!g.d_S) {
		BN_HIP(hipMalloc((void **)&g.d_S, sizeof(f128) * bn::kGroupMaxSlots));
		BN_HIP(hipMemsetAsync(g.d_S, 0, sizeof(f128) * bn::kGroupMaxSlots, ctx->stream));
	}
	return BN_OK;
}

void drop_predictions_touching(bn_ctx *ctx, const void *p, uint64_t n)
{
	for (auto &s : ctx->grp.sessions) {
		if (!s.pre_valid) continue;
		if (!((const char *)p < s.pre_hull_e && s.pre_hull_b < (const char *)p + n * sizeof(f128))) continue;
		for (uint32_t i = 0; i < s.m && s.pre_valid; i++)
			if (ranges_overlap(p, n, s.pre_lo[i], s.pre_row_len) || ranges_overlap(p, n, s.pre_hi[i], s.pre_row_len)) s.pre_valid = false;
	}
}

// what an evaluation request of the calculate_round_evals shape asks for
struct request {
	uint32_t m = 0, k = 0;
	uint64_t row_len = 0;
	const void *lo[kMaxArrays] = {}, *hi[kMaxArrays] = {};
	uint16_t pa[kMaxClaims] = {}, pb[kMaxClaims] = {};
	struct term {
		uint32_t value, claim, at_inf;
		f128 coeff;
	};
	std::vector<term> terms;
	uint32_t n_values = 0;
	f128 init[bn::kFinMaxValues] = {};
};

// Parses the op list.  false: not the shape (the single-claim dispatcher validates and answers it).
bool parse(const bn_memmap *maps, uint32_t n_maps, const bn_kop *ops, uint32_t n_ops, const uint32_t *ret_values, uint32_t n_ret, request &rq)
{
	if (n_ret == 0 || n_ret > (uint32_t)bn::kFinMaxRets) return false;
	std::vector<int> local_array(n_maps, -1); // Local buffer -> the array whose lo + hi it holds
	auto plain = [&](const bn_kslice &sl, const char *&p) {
		if (sl.buf >= n_maps || maps[sl.buf].kind == BN_MAP_LOCAL || !maps[sl.buf].d_data) return false;
		if (sl.off + sl.len > maps[sl.buf].len) return false;
		p = (const char *)maps[sl.buf].d_data + sl.off * sizeof(f128);
		return true;
	};
	// (a, b) -> claim.  The reference records the claims in ONE order, first at 1, then at infinity (v3/bivariate_product.rs:355-399):
	// a cursor that walks the claims found so far answers every look-up of the second half; whatever it misses goes through a
	// linear search (few claims) or a hash map built on the first miss (many) -- this parser runs in front of EVERY kernel launch,
	// the single-claim benchmark's included, and allocates nothing for it
	std::unordered_map<uint32_t, uint32_t> claim_index;
	bool claim_index_built = false;
	uint32_t cursor = 0;
	auto claim_of = [&](uint32_t a, uint32_t b) -> int {
		if (cursor < rq.k && rq.pa[cursor] == a && rq.pb[cursor] == b) return (int)cursor++;
		if (rq.k <= 16) {
			for (uint32_t c = 0; c < rq.k; c++)
				if (rq.pa[c] == a && rq.pb[c] == b) {
					cursor = c + 1;
					return (int)c;
				}
		} else {
			if (!claim_index_built) {
				claim_index.reserve(2 * (size_t)kMaxClaims);
				for (uint32_t c = 0; c < rq.k; c++) claim_index.emplace((uint32_t)rq.pa[c] << 16 | rq.pb[c], c);
				claim_index_built = true;
			}
			const auto it = claim_index.find(a << 16 | b);
			if (it != claim_index.end()) {
				cursor = it->second + 1;
				return (int)it->second;
			}
		}
		if (rq.k >= kMaxClaims) return -1;
		rq.pa[rq.k] = (uint16_t)a;
		rq.pb[rq.k] = (uint16_t)b;
		if (claim_index_built) claim_index.emplace(a << 16 | b, rq.k);
		cursor = 0; // (a new claim: the first half of the list is still being written)
		return (int)rq.k++;
	};
	// first pass: the arrays (every ADD of two mapped halves into a whole Local buffer) and the values
	for (uint32_t o = 0; o < n_ops; o++) {
		const bn_kop &op = ops[o];
		if (op.kind == BN_KOP_DECL_VALUE) {
			if (op.value >= (uint32_t)bn::kFinMaxValues) return false;
			if (op.value + 1 > rq.n_values) rq.n_values = op.value + 1;
			rq.init[op.value] = f128{op.scalar.lo, op.scalar.hi};
		} else if (op.kind == BN_KOP_ADD) {
			if (op.dst.buf >= n_maps || maps[op.dst.buf].kind != BN_MAP_LOCAL || maps[op.dst.buf].log_size > 40) return false;
			const uint64_t len = (uint64_t)1 << maps[op.dst.buf].log_size;
			if (op.dst.off != 0 || op.dst.len != len || op.src1.len != len || op.src2.len != len) return false;
			const char *p = nullptr, *q = nullptr;
			if (!plain(op.src1, p) || !plain(op.src2, q)) return false;
			if (local_array[op.dst.buf] >= 0 || rq.m >= kMaxArrays) return false; // (defined twice: not this shape)
			if (rq.m && len != rq.row_len) return false;
			rq.row_len = len;
			local_array[op.dst.buf] = (int)rq.m;
			rq.lo[rq.m] = p;
			rq.hi[rq.m] = q;
			rq.m++;
		} else if (op.kind != BN_KOP_SUM_COMPOSITION) {
			return false;
		}
	}
	if (rq.m == 0 || rq.row_len == 0) return false;
	// two arrays may not alias each other (a fold writes one while a job reads the other); and the way back from an upper half to
	// its array: a few arrays -- compare them all; many -- sorted by address
	std::vector<std::pair<const void *, uint32_t>> hi_sorted;
	if (rq.m <= 16) {
		for (uint32_t i = 0; i < rq.m; i++)
			for (uint32_t j = 0; j < i; j++)
				if (rq.lo[i] == rq.lo[j] || rq.hi[i] == rq.hi[j]) return false;
	} else {
		std::vector<const void *> lo_sorted(rq.lo, rq.lo + rq.m);
		std::sort(lo_sorted.begin(), lo_sorted.end());
		hi_sorted.reserve(rq.m);
		for (uint32_t i = 0; i < rq.m; i++) hi_sorted.push_back({rq.hi[i], i});
		std::sort(hi_sorted.begin(), hi_sorted.end());
		for (uint32_t i = 1; i < rq.m; i++)
			if (lo_sorted[i] == lo_sorted[i - 1] || hi_sorted[i].first == hi_sorted[i - 1].first) return false;
	}
	uint32_t hi_last[2] = {0, 0}; // (per factor: the array of the previous claim -- the next claim names the one after it, or the same)
	auto array_of_hi = [&](const void *p, int j) -> int {
		if (hi_last[j] + 1 < rq.m && rq.hi[hi_last[j] + 1] == p) return (int)hi_last[j] + 1;
		if (rq.hi[hi_last[j]] == p) return (int)hi_last[j];
		if (rq.m <= 16) {
			for (uint32_t i = 0; i < rq.m; i++)
				if (rq.hi[i] == p) return (int)i;
			return -1;
		}
		const auto it = std::lower_bound(hi_sorted.begin(), hi_sorted.end(), std::make_pair(p, (uint32_t)0));
		return it != hi_sorted.end() && it->first == p ? (int)it->second : -1;
	};
	// second pass, in order: a sum over Locals must come after the ADDs that define them
	std::vector<char> defined(n_maps, 0);
	for (uint32_t o = 0; o < n_ops; o++) {
		const bn_kop &op = ops[o];
		if (op.kind == BN_KOP_ADD) {
			defined[op.dst.buf] = 1;
			continue;
		}
		if (op.kind != BN_KOP_SUM_COMPOSITION) continue;
		if (!op.expr || op.expr->shape != bn_expr::PRODUCT || op.expr->product_vars.size() != 2 || op.value >= rq.n_values) return false;
		int arr[2];
		int n_local = 0;
		for (int j = 0; j < 2; j++) {
			const uint32_t v = op.expr->product_vars[j];
			if (v >= op.n_rows) return false;
			const bn_kslice &sl = op.rows[v];
			if (sl.buf >= n_maps || sl.len != rq.row_len) return false;
			if (maps[sl.buf].kind == BN_MAP_LOCAL) {
				if (sl.off != 0 || !defined[sl.buf] || local_array[sl.buf] < 0) return false;
				arr[j] = local_array[sl.buf];
				n_local++;
			} else {
				const char *p = nullptr;
				if (!plain(sl, p)) return false;
				arr[j] = array_of_hi(p, j); // the evaluation at 1 is the array's upper half
				if (arr[j] < 0) return false;
				hi_last[j] = (uint32_t)arr[j];
			}
		}
		if (n_local == 1) return false;
		const int c = claim_of((uint32_t)arr[0], (uint32_t)arr[1]);
		if (c < 0) return false;
		rq.terms.push_back(request::term{op.value, (uint32_t)c, n_local == 2 ? 1u : 0u, f128{op.scalar.lo, op.scalar.hi}});
	}
	for (uint32_t r = 0; r < n_ret; r++)
		if (ret_values[r] >= rq.n_values) return false;
	return rq.k > 0 && rq.k <= kMaxClaims; // (two accumulator slots per claim, bn::kGroupMaxSlots per launch)
}

void answer(const request &rq, const f128 *raw, const uint32_t *ret_values, uint32_t n_ret, bn_f128 *h_out)
{
	f128 vals[bn::kFinMaxValues];
	for (uint32_t v = 0; v < rq.n_values; v++) vals[v] = rq.init[v];
	for (const auto &t : rq.terms) {
		const f128 s = raw[2 * t.claim + t.at_inf];
		vals[t.value] ^= (t.coeff == f128{1, 0}) ? s : bn::mul_host(t.coeff, s);
	}
	for (uint32_t r = 0; r < n_ret; r++) h_out[r] = bn_f128{vals[ret_values[r]].lo, vals[ret_values[r]].hi};
}

// where array (lo, hi, row_len) comes out of a deferred fold: the batch and the index inside it.  The deferred folds are disjoint
// (group_defer_fold), so an address is the output of at most one of them and the input of at most one: two hash maps, built once
// per evaluation (a prover of the keccak width names > 100 arrays per call).
struct fold_ref {
	int f = -1, j = -1;
};
struct fold_index {
	std::unordered_map<const void *, fold_ref> by_out, by_in;
	bool built = false;
	fold_ref hint; // where the previous look-up was answered: the next array of a prover is the next entry of the same batch, as a rule
	void reset()
	{
		built = false;
		hint = fold_ref{};
	}
	void build(const bn_ctx *ctx)
	{
		by_out.clear();
		by_in.clear();
		size_t total = 0;
		for (const auto &g : ctx->grp.folds) total += g.count;
		by_out.reserve(2 * total);
		by_in.reserve(2 * total);
		for (size_t f = 0; f < ctx->grp.folds.size(); f++) {
			const auto &g = ctx->grp.folds[f];
			for (uint32_t j = 0; j < g.count; j++) {
				by_out.emplace(g.x0[j], fold_ref{(int)f, (int)j});
				by_in.emplace(g.src0[j], fold_ref{(int)f, (int)j});
			}
		}
		built = true;
	}
	template <class Match>
	bool try_hint(const bn_ctx *ctx, Match &&match, fold_ref &out)
	{
		const auto &folds = ctx->grp.folds;
		for (int step = 1; step >= 0; step--) { // the entry after the last answer, then the first entry of the next batch
			fold_ref c = hint;
			if (c.f < 0) c = fold_ref{0, -1};
			if (step) {
				c.j++;
			} else {
				c.f++;
				c.j = 0;
			}
			if (c.f < (int)folds.size() && c.j < (int)folds[c.f].count && match(folds[c.f], (uint32_t)c.j)) {
				out = hint = c;
				return true;
			}
		}
		return false;
	}
	fold_ref output(const bn_ctx *ctx, const void *lo, const void *hi, uint64_t row_len)
	{
		if (ctx->grp.folds.empty()) return fold_ref{};
		fold_ref r;
		if (try_hint(ctx, [&](const bn_ctx::group_fold &g, uint32_t j) { return g.x0[j] == lo && g.n == 2 * row_len && (const void *)at(lo, row_len) == hi; }, r)) return r;
		if (!built) build(ctx);
		const auto it = by_out.find(lo);
		if (it == by_out.end() || ctx->grp.folds[it->second.f].n != 2 * row_len || (const void *)at(lo, row_len) != hi) return fold_ref{};
		return hint = it->second;
	}
	fold_ref input(const bn_ctx *ctx, const void *lo, const void *hi, uint64_t row_len)
	{
		if (ctx->grp.folds.empty()) return fold_ref{};
		fold_ref r;
		if (try_hint(ctx, [&](const bn_ctx::group_fold &g, uint32_t j) { return g.src0[j] == lo && g.n == row_len && g.x1[j] == hi; }, r)) return r;
		if (!built) build(ctx);
		const auto it = by_in.find(lo);
		if (it == by_in.end()) return fold_ref{};
		const auto &g = ctx->grp.folds[it->second.f];
		if (g.n != row_len || g.x1[it->second.j] != hi) return fold_ref{};
		return hint = it->second;
	}
};

// the jobs of one prover's evaluation: arrays (lo, hi) of row_len points each, array i coming out of deferred fold ref[i]
// (f < 0: it is up to date in memory); appends the jobs (slot = slot0 + 2 * claim).
//
//   * a claim whose two arrays both wait for their fold (one challenge) and are not yet folded by another job: fold + evaluate
//     (kind 0) -- a matching of the claim graph, the claims over arrays of degree one first (a prover with disjoint claims: all);
//   * a claim ONE of whose arrays still waits while the other is folded by a kind-0 / kind-3 job or is up to date: fold that array
//     and evaluate against the other as it is (kind 4) -- piop::prove's shape, every committed multilinear of a size against the
//     few ring-switch transparents of that size (piop/prove.rs:262-287): one kind-0 job per transparent, a kind-4 job for
//     every other committed column;
//   * the arrays still waiting after that: fold only (kind 3), two per job;
//   * every other claim: evaluate (kind 1);
//   * where a job reads an array that a job of THIS launch folds, the jobs concerned form a CHAIN -- folding jobs, then the
//     kind-4 jobs, then the evaluating ones: the same workgroups run them on the same tiles, so that a workgroup only ever reads
//     back what it has written itself (kernels_group.hip; `acquire` on the first job of the second and of the third part).
//     Everything else stays a job of its own.
//
// false (nothing appended): more than `room` jobs -- the caller falls back to plan_prefold.
bool plan_chained(const bn_ctx *ctx, uint32_t m, uint32_t k, const uint16_t *pa, const uint16_t *pb, const void *const *lo, const void *const *hi, uint64_t row_len,
                  const fold_ref *ref, uint32_t slot0, size_t room, std::vector<bn::group_job> &jobs, uint64_t &n_fused, uint64_t &n_fold_only, uint64_t &n_chains)
{
	const auto &folds = ctx->grp.folds;
	std::vector<uint32_t> deg(m, 0);
	for (uint32_t c = 0; c < k; c++) {
		deg[pa[c]]++;
		deg[pb[c]]++;
	}
	std::vector<char> matched(m, 0), fused_claim(k, 0);
	auto fusable = [&](uint32_t c) {
		const uint32_t a = pa[c], b = pb[c];
		return a != b && !matched[a] && !matched[b] && ref[a].f >= 0 && ref[b].f >= 0 && folds[ref[a].f].z == folds[ref[b].f].z;
	};
	for (int pass = 0; pass < 2; pass++)
		for (uint32_t c = 0; c < k; c++) {
			if (fused_claim[c] || !fusable(c)) continue;
			if (pass == 0 && (deg[pa[c]] != 1 || deg[pb[c]] != 1)) continue;
			fused_claim[c] = 1;
			matched[pa[c]] = matched[pb[c]] = 1;
		}
	// half[c]: 0 = no, 1 = the claim folds pa[c] and reads pb[c] as it is, 2 = the other way round; halved[i]: array i is folded by
	// such a job.  The array read as it is must be final when the kind-4 jobs run: matched (folded by a kind-0 job, in front) or
	// up to date -- neither is ever folded by a kind-4 job itself.  (After the matching every claim that is left has a matched
	// or up-to-date array unless its two folds differ in their challenge.)
	std::vector<char> half(k, 0), halved(m, 0);
	for (uint32_t c = 0; c < k; c++) {
		if (fused_claim[c] || pa[c] == pb[c]) continue;
		for (int side = 0; side < 2 && !half[c]; side++) {
			const uint32_t a = side ? pb[c] : pa[c], b = side ? pa[c] : pb[c];
			if (ref[a].f < 0 || matched[a] || halved[a]) continue;
			if (!(matched[b] || ref[b].f < 0)) continue;
			half[c] = (char)(1 + side);
			halved[a] = 1;
		}
	}
	// read_back[i]: array i is folded by this launch and read as it is by a kind-4 or a kind-1 job of it
	std::vector<char> read_back(m, 0);
	for (uint32_t c = 0; c < k; c++) {
		if (fused_claim[c]) continue;
		if (half[c]) {
			const uint32_t b = half[c] == 1 ? pb[c] : pa[c];
			if (ref[b].f >= 0) read_back[b] = 1;
			continue;
		}
		for (uint32_t i : {(uint32_t)pa[c], (uint32_t)pb[c]})
			if (ref[i].f >= 0) read_back[i] = 1;
	}
	std::vector<uint32_t> left[2]; // arrays to fold only: [1] = read back (chain), [0] = not
	for (uint32_t i = 0; i < m; i++)
		if (ref[i].f >= 0 && !matched[i] && !halved[i]) left[read_back[i] ? 1 : 0].push_back(i);
	auto fold_only_jobs = [&](const std::vector<uint32_t> &v) {
		// (two arrays per job where they share the challenge)
		size_t n = 0;
		for (size_t q = 0; q < v.size();) {
			const bool two = q + 1 < v.size() && folds[ref[v[q]].f].z == folds[ref[v[q + 1]].f].z;
			q += two ? 2 : 1;
			n++;
		}
		return n;
	};
	if ((size_t)k + fold_only_jobs(left[0]) + fold_only_jobs(left[1]) > room) return false;
	auto fold_side = [&](bn::group_job &j, int sd, uint32_t i) {
		const auto &f = folds[ref[i].f];
		j.x0[sd] = f.src0[ref[i].j];
		j.x1[sd] = f.x1[ref[i].j];
		j.out[sd] = f.x0[ref[i].j];
		j.z = f.z;
	};
	std::vector<bn::group_job> chain_w, chain_h, chain_e, alone;
	for (uint32_t c = 0; c < k; c++) {
		if (!fused_claim[c]) continue;
		bn::group_job j{};
		j.kind = 0;
		j.n = row_len;
		j.slot = slot0 + 2 * c;
		fold_side(j, 0, pa[c]);
		fold_side(j, 1, pb[c]);
		(read_back[pa[c]] || read_back[pb[c]] ? chain_w : alone).push_back(j);
		n_fused++;
	}
	for (int rb = 0; rb < 2; rb++)
		for (size_t q = 0; q < left[rb].size();) {
			bn::group_job j{};
			j.kind = 3;
			j.n = row_len;
			fold_side(j, 0, left[rb][q]);
			const bool two = q + 1 < left[rb].size() && folds[ref[left[rb][q]].f].z == folds[ref[left[rb][q + 1]].f].z;
			if (two) fold_side(j, 1, left[rb][q + 1]);
			q += two ? 2 : 1;
			(rb ? chain_w : alone).push_back(j);
			n_fold_only++;
		}
	for (uint32_t c = 0; c < k; c++) {
		if (!half[c]) continue;
		const uint32_t a = half[c] == 1 ? pa[c] : pb[c], b = half[c] == 1 ? pb[c] : pa[c];
		bn::group_job j{};
		j.kind = 4;
		j.n = row_len;
		j.slot = slot0 + 2 * c;
		fold_side(j, 0, a);
		j.x0[1] = lo[b];
		j.x1[1] = hi[b];
		(ref[b].f >= 0 || read_back[a] ? chain_h : alone).push_back(j);
		n_fused++;
	}
	for (uint32_t c = 0; c < k; c++) {
		if (fused_claim[c] || half[c]) continue;
		const uint32_t a = pa[c], b = pb[c];
		bn::group_job j{};
		j.kind = 1;
		j.n = row_len;
		j.slot = slot0 + 2 * c;
		j.x0[0] = lo[a];
		j.x1[0] = hi[a];
		j.x0[1] = lo[b];
		j.x1[1] = hi[b];
		(ref[a].f >= 0 || ref[b].f >= 0 ? chain_e : alone).push_back(j);
	}
	if (!chain_w.empty() || !chain_h.empty() || !chain_e.empty()) {
		// (a reading part's first job waits for the workgroup's own stores of everything in front of it -- unless nothing is)
		if (!chain_h.empty() && !chain_w.empty()) chain_h[0].acquire = 1;
		if (!chain_e.empty() && (!chain_w.empty() || !chain_h.empty())) chain_e[0].acquire = 1;
		const size_t at = jobs.size();
		jobs.insert(jobs.end(), chain_w.begin(), chain_w.end());
		jobs.insert(jobs.end(), chain_h.begin(), chain_h.end());
		jobs.insert(jobs.end(), chain_e.begin(), chain_e.end());
		jobs[at].chain = (uint32_t)(jobs.size() - at) - 1;
		if (jobs[at].chain) n_chains++;
	}
	jobs.insert(jobs.end(), alone.begin(), alone.end());
	return true;
}

// without chains (arrays below the size where chains pay, or more jobs than a launch carries): claims over arrays of degree one
// are fused (kind 0); a claim over an array of degree one and a shared one folds the former on the way (kind 4) -- the shared
// array, like every other waiting array, is folded by a plain launch in front (prefold)
void plan_prefold(const bn_ctx *ctx, uint32_t m, uint32_t k, const uint16_t *pa, const uint16_t *pb, const void *const *lo, const void *const *hi, uint64_t row_len,
                  const fold_ref *ref, uint32_t slot0, std::vector<bn::group_job> &jobs, std::vector<char> &prefold /*[m]*/, uint64_t &n_fused)
{
	std::vector<uint32_t> deg(m, 0);
	for (uint32_t c = 0; c < k; c++) {
		deg[pa[c]]++;
		deg[pb[c]]++;
	}
	std::vector<char> fused_arr(m, 0);
	for (uint32_t c = 0; c < k; c++) {
		const uint32_t a = pa[c], b = pb[c];
		bn::group_job j{};
		j.n = row_len;
		j.slot = slot0 + 2 * c;
		auto waits_alone = [&](uint32_t i) { return deg[i] == 1 && ref[i].f >= 0; };
		if (a != b && waits_alone(a) && waits_alone(b) && ctx->grp.folds[ref[a].f].z == ctx->grp.folds[ref[b].f].z) {
			const auto &fa = ctx->grp.folds[ref[a].f], &fb = ctx->grp.folds[ref[b].f];
			j.kind = 0;
			j.z = fa.z;
			j.x0[0] = fa.src0[ref[a].j];
			j.x1[0] = fa.x1[ref[a].j];
			j.out[0] = fa.x0[ref[a].j];
			j.x0[1] = fb.src0[ref[b].j];
			j.x1[1] = fb.x1[ref[b].j];
			j.out[1] = fb.x0[ref[b].j];
			fused_arr[a] = fused_arr[b] = 1;
			n_fused++;
		} else if (a != b && (waits_alone(a) || waits_alone(b))) {
			// (the other array is shared, or up to date, or waits under another challenge: final by the time this launch runs)
			const uint32_t own = waits_alone(a) ? a : b, other = own == a ? b : a;
			const auto &fo = ctx->grp.folds[ref[own].f];
			j.kind = 4;
			j.z = fo.z;
			j.x0[0] = fo.src0[ref[own].j];
			j.x1[0] = fo.x1[ref[own].j];
			j.out[0] = fo.x0[ref[own].j];
			j.x0[1] = lo[other];
			j.x1[1] = hi[other];
			fused_arr[own] = 1;
			n_fused++;
		} else {
			j.kind = 1;
			j.x0[0] = lo[a];
			j.x1[0] = hi[a];
			j.x0[1] = lo[b];
			j.x1[1] = hi[b];
		}
		jobs.push_back(j);
	}
	for (uint32_t i = 0; i < m; i++) prefold[i] = (ref[i].f >= 0 && !fused_arr[i]) ? 1 : 0;
}

int unhost(bn_ctx *ctx, bn_ctx::group_session &s);

bn_ctx::group_session *session_for(bn_ctx *ctx, const request &rq, const fold_ref *ref, int *rc_out)
{
	auto &ss = ctx->grp.sessions;
	auto same_claims = [&](const bn_ctx::group_session &s) {
		if (s.m != rq.m || s.k != rq.k) return false;
		for (uint32_t c = 0; c < rq.k; c++)
			if (s.pa[c] != rq.pa[c] || s.pb[c] != rq.pb[c]) return false;
		return true;
	};
	for (auto &s : ss) {
		if (!same_claims(s)) continue;
		bool cont = true, same = s.row_len == rq.row_len;
		for (uint32_t i = 0; i < rq.m; i++) {
			if (ref[i].f < 0) {
				cont = false;
			} else {
				const auto &g = ctx->grp.folds[ref[i].f];
				if (g.src0[ref[i].j] != s.lo[i] || g.x1[ref[i].j] != s.hi[i] || g.n != s.row_len) cont = false;
			}
			if (s.lo[i] != rq.lo[i] || s.hi[i] != rq.hi[i]) same = false;
		}
		if (cont || same) return &s;
	}
	if (ss.size() >= kMaxSessions) {
		// the least recently used description goes -- never one the host still owes a write-back for (its folded copies are the only
		// place the caller's folds exist) unless every session is such a one, and then only after the write-back
		size_t old = ss.size();
		for (size_t i = 0; i < ss.size(); i++)
			if (!ss[i].hosted && (old == ss.size() || ss[i].stamp < ss[old].stamp)) old = i;
		if (old == ss.size()) {
			old = 0;
			for (size_t i = 1; i < ss.size(); i++)
				if (ss[i].stamp < ss[old].stamp) old = i;
			const int rc = unhost(ctx, ss[old]);
			if (rc) {
				*rc_out = rc;
				return nullptr;
			}
		}
		ss.erase(ss.begin() + (long)old);
	}
	ss.emplace_back();
	return &ss.back();
}

void record(bn_ctx *ctx, bn_ctx::group_session &s, const request &rq)
{
	s.m = rq.m;
	s.k = rq.k;
	s.row_len = rq.row_len;
	s.lo.assign(rq.lo, rq.lo + rq.m);
	s.hi.assign(rq.hi, rq.hi + rq.m);
	s.pa.assign(rq.pa, rq.pa + rq.k);
	s.pb.assign(rq.pb, rq.pb + rq.k);
	s.pre_valid = false;
	s.stamp = ++ctx->grp.stamp;
}

// ---- hosted sessions: the last rounds of a prover as host arithmetic ------------------------------------------------------------
// Once every array of a prover is at most grp.ht_max elements, ONE launch hands them to the host (k_group_mirror: the prover's
// deferred fold on the way, the basis change to the power basis of hostmul_clmul.cpp) and from then on its execute() and fold()
// calls are host arithmetic on the host's copies -- no launch, no round trip -- as long as they are the expected ones (the
// evaluation of exactly the current halves, the fold of exactly the current arrays, in place after the first).  The host folds
// in place exactly as the device would, so the first h_n0 elements of its copies ARE the caller's buffers after the folds:
// whoever looks at that memory first (a read, a copy, a foreign kernel: group_flush / group_flush_touching) triggers one
// write-back launch.  Reads of elements of the current arrays (finish()) are answered from the copies without it.
// does an access to [p, p + n) conflict with hosted session s?  A write into its current arrays ends the hosting (the host's copy
// is stale); any access to memory its pending write-back covers needs that write-back first; a READ of the current arrays while
// nothing is pending (the device still holds what the host holds) conflicts with nothing.
bool session_touches(const bn_ctx::group_session &s, const void *p, uint64_t n, bool write)
{
	if (!s.hosted) return false;
	for (uint32_t j = 0; j < s.m; j++) {
		const uint64_t half = s.h_len > 1 ? s.h_len / 2 : 1;
		if (write && (ranges_overlap(p, n, s.h_lo[j], half) || (s.h_len > 1 && ranges_overlap(p, n, s.h_hi[j], half)))) return true;
		if (s.h_levels && ranges_overlap(p, n, s.h_out[j], s.h_n0)) return true;
	}
	return false;
}

int stage_alloc(bn_ctx *ctx)
{
	auto &g = ctx->grp;
	if (g.h_stage) return BN_OK;
	if (hipHostMalloc(&g.h_stage, 2 * bn::kGroupTailMaxElems * sizeof(f128), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
		(void)hipGetLastError();
		g.h_stage = nullptr;
		return BN_ERR_ALLOC;
	}
	BN_HIP(hipHostGetDevicePointer(&g.d_stage, g.h_stage, 0));
	return BN_OK;
}

// the device catches up with the folds the host performed for session s; the session is an ordinary one again
int unhost(bn_ctx *ctx, bn_ctx::group_session &s)
{
	if (!s.hosted) return BN_OK;
	if (s.h_levels) {
		// (the staging's write-back half may still be read by the previous write-back's kernel: one launch at a time)
		BN_HIP(hipStreamSynchronize(ctx->stream));
		uint64_t *stg = (uint64_t *)ctx->grp.h_stage + 2 * bn::kGroupTailMaxElems;
		bn::group_writeback_args a{};
		a.count = s.m;
		a.n0 = (uint32_t)s.h_n0;
		// (the stream is idle: neither the staging's write-back half nor the write-back's pointer table is being read)
		bn::group_tables *tb = (bn::group_tables *)ctx->grp.h_tables;
		for (uint32_t j = 0; j < s.m; j++) {
			std::memcpy(stg + 2 * (size_t)s.h_n0 * j, s.hy[j].data(), (size_t)s.h_n0 * 16);
			tb->writeback.out[j] = s.h_out[j];
		}
		a.ptrs = &((const bn::group_tables *)ctx->grp.d_tables)->writeback;
		a.staging = (const f128 *)((const char *)ctx->grp.d_stage + bn::kGroupTailMaxElems * sizeof(f128));
		a.phi_inv = (const uint4 *)((const char *)ctx->d_phi + 512 * sizeof(f128));
		__atomic_thread_fence(__ATOMIC_SEQ_CST);
		prof_scope ps(ctx, BN_PROF_FOLD);
		BN_HIP(bn::launch_group_writeback(ctx->stream, a));
		ctx->grp.hosted_writebacks++;
	}
	s.hosted = false;
	s.h_levels = 0;
	s.hy.clear();
	return BN_OK;
}

int unhost_touching(bn_ctx *ctx, const void *p, uint64_t n, bool write)
{
	for (auto &s : ctx->grp.sessions)
		if (session_touches(s, p, n, write)) {
			const int rc = unhost(ctx, s);
			if (rc) return rc;
		}
	return BN_OK;
}

int unhost_all(bn_ctx *ctx)
{
	for (auto &s : ctx->grp.sessions) {
		const int rc = unhost(ctx, s);
		if (rc) return rc;
	}
	return BN_OK;
}

void host_answer(bn_ctx *ctx, const bn_ctx::group_session &s, const request &rq, const uint32_t *ret_values, uint32_t n_ret, bn_f128 *h_out)
{
	std::vector<f128> raw(2 * (size_t)rq.k);
	const size_t half = (size_t)(s.h_len / 2);
	for (uint32_t c = 0; c < rq.k; c++) {
		bn::hp128 y1, yi;
		bn::hostpoly_round_sums(reinterpret_cast<const bn::hp128 *>(s.hy[rq.pa[c]].data()), reinterpret_cast<const bn::hp128 *>(s.hy[rq.pb[c]].data()), half, &y1, &yi);
		raw[2 * c] = bn::hostpoly_to_tower(y1);
		raw[2 * c + 1] = bn::hostpoly_to_tower(yi);
	}
	answer(rq, raw.data(), ret_values, n_ret, h_out);
	ctx->grp.hosted_evals++;
	ctx->grp.evals++;
}

int wait_mail(bn_ctx *ctx, uint64_t seq)
{
	volatile uint64_t *seqw = &ctx->h_mail[64].lo;
	uint64_t spins = 0;
	while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) {
		if (++spins > (1ull << 22)) {
			BN_HIP(hipStreamSynchronize(ctx->stream));
			if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
			break;
		}
	}
	return BN_OK;
}
} // namespace

int group_res_alloc(bn_ctx *ctx) { return res_alloc(ctx); }

bool legacy_state_active(const bn_ctx *ctx)
{
	return ctx->pend.active || ctx->pend2.active || ctx->arm.active || ctx->tail.active || ctx->ht.active || ctx->shadow.valid || ctx->pre.valid ||
	       ctx->side_busy || !ctx->side_queue.empty();
}

bool group_fold_applies(const bn_ctx *ctx, uint32_t count, uint32_t scale_mask)
{
	const auto &g = ctx->grp;
	if (!g.enabled || !ctx->lazy_fold || ctx->peer.active || scale_mask || ctx->tail_max_n_in || count == 0 || count > kMaxArrays) return false;
	// (a lone batch of two arrays with nothing else waiting is the single-claim shape: the armed / two-round / host-tail machinery
	// of abi.cpp keeps it)
	return g.on || !g.folds.empty() || count != 2;
}

// `on` (single-claim calls join the group path) lasts as long as something of the group path is alive: it ends with a full flush,
// and with a selective one that leaves no deferred fold, no hosted prover and no sums computed ahead -- a multi-claim prover that
// has finished does not pull the single-claim provers after it off their armed / two-round / host-tail machinery
static void group_settle(bn_ctx *ctx)
{
	auto &g = ctx->grp;
	if (!g.on || !g.folds.empty()) return;
	for (const auto &s : g.sessions)
		if ((s.hosted && s.h_len > 1) || s.pre_valid) return; // (a hosted prover folded down to one element has finished: only its write-back is owed)
	g.on = false;
}

bool group_independent(const bn_ctx *ctx, const void *p, uint64_t n, bool write)
{
	for (const auto &f : ctx->grp.folds)
		if (fold_touches(f, p, n)) return false;
	for (const auto &s : ctx->grp.sessions)
		if (session_touches(s, p, n, write)) return false;
	return true;
}

// Does a fold batch overlap ITSELF: an output against any other range of the batch (another array's output, lower or upper half; its
// own upper half; its own lower half when the fold is out of place)?  One sweep over the ranges sorted by address: a range overlaps
// an earlier one exactly when it starts before the furthest end seen so far (all ranges are n elements long).
static bool batch_overlaps_itself(void *const *x0, const void *const *src0, const void *const *x1, uint32_t count, uint64_t n)
{
	struct iv {
		const char *b;
		bool write;
	};
	iv small[96];
	std::vector<iv> big;
	iv *v = small;
	if (3 * (size_t)count > sizeof(small) / sizeof(small[0])) {
		big.resize(3 * (size_t)count);
		v = big.data();
	}
	size_t cnt = 0;
	for (uint32_t i = 0; i < count; i++) {
		v[cnt++] = iv{(const char *)x0[i], true};
		v[cnt++] = iv{(const char *)x1[i], false};
		if (src0[i] != x0[i]) v[cnt++] = iv{(const char *)src0[i], false};
	}
	const auto less = [](const iv &a, const iv &b) { return a.b < b.b; };
	if (!std::is_sorted(v, v + cnt, less)) std::sort(v, v + cnt, less);
	const char *end_any = nullptr, *end_write = nullptr;
	const size_t bytes = (size_t)n * sizeof(f128);
	for (size_t q = 0; q < cnt; q++) {
		const iv &r = v[q];
		if (r.write ? (end_any && r.b < end_any) : (end_write && r.b < end_write)) return true;
		if (!end_any || r.b + bytes > end_any) end_any = r.b + bytes;
		if (r.write && (!end_write || r.b + bytes > end_write)) end_write = r.b + bytes;
	}
	return false;
}

// The fold batch is the fold of a hosted prover's current arrays: performed on the host's copies (true), or not the expected call.
static bool host_fold(bn_ctx *ctx, void *const *x0, const void *const *src0, const void *const *x1, uint32_t count, uint64_t n, f128 z)
{
	for (auto &s : ctx->grp.sessions) {
		if (!s.hosted || s.m != count || s.h_len != 2 * n) continue;
		std::vector<int> of(count, -1); // batch index -> session array
		std::vector<char> used(s.m, 0);
		bool ok = true;
		for (uint32_t i = 0; i < count && ok; i++) {
			for (uint32_t j = 0; j < s.m; j++)
				if (!used[j] && src0[i] == s.h_lo[j] && x1[i] == s.h_hi[j]) {
					of[i] = (int)j;
					used[j] = 1;
					break;
				}
			ok = of[i] >= 0;
		}
		if (!ok) continue;
		// later folds are in place on the previous output (what the write-back of the host copies assumes); no output may overlap
		// what the batch reads of ANOTHER array or the upper half of its own
		for (uint32_t i = 0; i < count && ok; i++)
			if (s.h_levels > 0 && x0[i] != src0[i]) ok = false;
		ok = ok && !batch_overlaps_itself(x0, src0, x1, count, n); // (a sweep: a hundred arrays were 30 000 pairwise checks per round)
		if (!ok) return false; // (the caller's general path writes the copies back and folds on the device)
		if (s.h_levels == 0) {
			s.h_n0 = n;
			s.h_out.assign(s.m, nullptr);
			for (uint32_t i = 0; i < count; i++) s.h_out[of[i]] = x0[i];
		}
		const bn::hp128 pz = bn::hostpoly_from_tower(z);
		for (uint32_t i = 0; i < count; i++) {
			const int j = of[i];
			bn::hostpoly_fold(reinterpret_cast<bn::hp128 *>(s.hy[j].data()), (size_t)n, pz);
			s.h_lo[j] = x0[i];
			s.h_hi[j] = at(x0[i], n / 2);
		}
		s.h_len = n;
		s.h_levels++;
		// (the session's description follows the arrays: what a later evaluation will name)
		s.row_len = n / 2;
		for (uint32_t j = 0; j < s.m; j++) {
			s.lo[j] = s.h_lo[j];
			s.hi[j] = s.h_hi[j];
		}
		ctx->grp.hosted_folds++;
		ctx->grp.on = true;
		return true;
	}
	return false;
}

// A host read of [p, p + n) that lies inside ONE current array of a hosted prover: answered from the host's copy (true).
bool group_host_read(bn_ctx *ctx, const void *p, uint64_t n, bn_f128 *h_dst)
{
	for (const auto &s : ctx->grp.sessions) {
		if (!s.hosted || n == 0) continue;
		for (uint32_t j = 0; j < s.m; j++) {
			if (s.h_len > 1 && s.h_hi[j] != (const void *)at(s.h_lo[j], s.h_len / 2)) continue; // (halves not adjacent: no single range)
			const char *b0 = (const char *)s.h_lo[j], *b1 = b0 + s.h_len * sizeof(f128);
			if ((const char *)p < b0 || (const char *)p + n * sizeof(f128) > b1) continue;
			const size_t off = (size_t)((const char *)p - b0) / sizeof(f128);
			for (uint64_t e = 0; e < n; e++) {
				const f128 v = bn::hostpoly_to_tower(bn::hp128{s.hy[j][2 * (off + e)], s.hy[j][2 * (off + e) + 1]});
				h_dst[e] = bn_f128{v.lo, v.hi};
			}
			group_settle(ctx); // (finish() of a hosted prover: with its last read nothing of the group path may be alive any more)
			return true;
		}
	}
	return false;
}

// the single-claim state gives way: a plain deferred fold moves over as it is, everything else of it is flushed
static int legacy_to_group(bn_ctx *ctx)
{
	if (!legacy_state_active(ctx)) return BN_OK;
	bn_ctx::group_fold moved;
	bool have = false;
	if (ctx->pend.active && !ctx->pend2.active && !ctx->pend.scale_mask && !ctx->ht.active && !ctx->tail.active && !(ctx->shadow.valid && ctx->shadow.fold_pending) &&
	    ctx->pend.count <= kMaxArrays) {
		const auto &pf = ctx->pend;
		moved.n = pf.n;
		moved.z = pf.z;
		for (uint32_t i = 0; i < pf.count; i++) moved.push(pf.x0[i], pf.x1[i], pf.src0[i]);
		ctx->pend.active = false;
		have = true;
	}
	const int rc = flush_legacy(ctx);
	if (rc) return rc;
	if (have) ctx->grp.folds.push_back(moved);
	return BN_OK;
}

int group_defer_fold(bn_ctx *ctx, void *const *x0, const void *const *src0, const void *const *x1, uint32_t count, uint64_t n, f128 z)
{
	struct lap_on_exit {
		phase_clock pc;
		~lap_on_exit() { pc.lap(bn_ctx::group_state::P_DEFER); }
	} timer{phase_clock(ctx->grp)};
	int rc = legacy_to_group(ctx);
	if (rc) return rc;
	auto &g = ctx->grp;
	if (host_fold(ctx, x0, src0, x1, count, n, z)) { // (a hosted prover's fold: performed on the host's copies)
		timer.pc.lap(bn_ctx::group_state::P_HOST_FOLD);
		return BN_OK;
	}
	// a batch whose arrays overlap what a waiting batch reads or writes is ordered behind it: the waiting ones run first
	bool clash = false;
	for (uint32_t i = 0; i < count && !clash; i++)
		clash = !group_independent(ctx, x0[i], n, true) || !group_independent(ctx, x1[i], n, false) || !group_independent(ctx, src0[i], n, false);
	// ... and so is one that overlaps ITSELF: an output against any other range of the batch (the jobs of a launch run concurrently;
	// an out-of-place fold whose output overlaps its own inputs).  One sweep over the ranges sorted by address: a range overlaps
	// an earlier one exactly when it starts before the furthest end seen so far.
	if (!clash) clash = batch_overlaps_itself(x0, src0, x1, count, n);
	bn_ctx::group_fold f;
	f.n = n;
	f.z = z;
	for (uint32_t i = 0; i < count; i++) f.push(x0[i], x1[i], src0[i]);
	if (clash) {
		rc = group_flush(ctx);
		if (rc) return rc;
		return launch_fold(ctx, f);
	}
	// the sums computed ahead describe arrays BEFORE this fold: a prover that folds without having asked for them has moved on
	for (uint32_t i = 0; i < count; i++) drop_predictions_touching(ctx, x0[i], n);
	g.folds.push_back(std::move(f));
	g.on = true;
	return BN_OK;
}

int group_flush(bn_ctx *ctx)
{
	auto &g = ctx->grp;
	g.on = false;
	for (auto &s : g.sessions) s.pre_valid = false;
	{
		const int rc = unhost_all(ctx);
		if (rc) return rc;
	}
	if (g.folds.empty()) return BN_OK;
	std::vector<bn_ctx::group_fold> todo;
	todo.swap(g.folds);
	for (const auto &f : todo) {
		const int rc = launch_fold(ctx, f);
		if (rc) return rc;
	}
	return BN_OK;
}

int group_flush_touching(bn_ctx *ctx, const void *p, uint64_t n, bool publish_tiny, bool write)
{
	auto &g = ctx->grp;
	if (write) drop_predictions_touching(ctx, p, n);
	{
		const int rc = unhost_touching(ctx, p, n, write);
		if (rc) return rc;
	}
	for (size_t i = 0; i < g.folds.size();) {
		if (fold_touches(g.folds[i], p, n)) {
			const bn_ctx::group_fold f = g.folds[i];
			g.folds.erase(g.folds.begin() + (long)i);
			if (publish_tiny && (uint64_t)f.count * f.n <= 64 && f.count <= (uint32_t)bn::kFoldBatchMax && !ctx->mirror.valid) {
				// the caller is a host read of a handful of elements (finish(): one copy_d2h per multilinear, each a stream
				// synchronisation otherwise): fold and mirror the results into the mailbox in one launch; the reads that follow are
				// served from there (bn_copy_d2h)
				const uint64_t seq = ++ctx->mail_seq;
				prof_scope ps(ctx, BN_PROF_FOLD);
				BN_HIP(bn::launch_fold_publish(ctx->stream, f.x0.data(), f.src0.data(), f.x1.data(), f.count, (uint32_t)f.n, f.z, ctx->d_mail, seq));
				ctx->grp.flushed_folds++;
				ctx->mirror.valid = true;
				ctx->mirror.host = false;
				ctx->mirror.seq = seq;
				ctx->mirror.count = f.count;
				ctx->mirror.n = (uint32_t)f.n;
				for (uint32_t q = 0; q < f.count; q++) ctx->mirror.ptr[q] = f.x0[q];
				continue;
			}
			const int rc = launch_fold(ctx, f);
			if (rc) return rc;
		} else {
			i++;
		}
	}
	group_settle(ctx);
	return BN_OK;
}

void group_note_write(bn_ctx *ctx, const void *p, uint64_t n) { drop_predictions_touching(ctx, p, n); }

int group_eval(bn_ctx *ctx, const bn_memmap *maps, uint32_t n_maps, const bn_kop *ops, uint32_t n_ops, const uint32_t *ret_values, uint32_t n_ret,
               bn_f128 *h_out, bool *handled)
{
	*handled = false;
	auto &g = ctx->grp;
	if (!g.enabled || !ctx->lazy_fold || ctx->peer.active || ctx->tail_max_n_in || !h_out) return BN_OK;
	phase_clock pc(g);
	request rq;
	if (!parse(maps, n_maps, ops, n_ops, ret_values, n_ret, rq)) return BN_OK;
	pc.lap(bn_ctx::group_state::P_PARSE);
	// ---- a hosted prover's evaluation of exactly its current halves: host arithmetic
	for (auto &s : g.sessions) {
		if (!s.hosted || s.m != rq.m || s.k != rq.k || s.h_len != 2 * rq.row_len) continue;
		bool same = true;
		for (uint32_t i = 0; i < rq.m && same; i++) same = s.h_lo[i] == rq.lo[i] && s.h_hi[i] == rq.hi[i];
		for (uint32_t c = 0; c < rq.k && same; c++) same = s.pa[c] == rq.pa[c] && s.pb[c] == rq.pb[c];
		if (!same) continue;
		host_answer(ctx, s, rq, ret_values, n_ret, h_out);
		s.stamp = ++g.stamp;
		*handled = true;
		pc.lap(bn_ctx::group_state::P_HOSTED);
		return BN_OK;
	}
	// ---- sums computed ahead for exactly this request?
	for (auto &s : g.sessions) {
		if (!s.pre_valid || s.m != rq.m || s.k != rq.k || s.pre_row_len != rq.row_len) continue;
		bool same = true;
		for (uint32_t i = 0; i < rq.m && same; i++) same = s.pre_lo[i] == rq.lo[i] && s.pre_hi[i] == rq.hi[i];
		for (uint32_t c = 0; c < rq.k && same; c++) same = s.pa[c] == rq.pa[c] && s.pb[c] == rq.pb[c];
		if (!same) continue;
		answer(rq, s.pre_raw.data(), ret_values, n_ret, h_out);
		record(ctx, s, rq);
		g.spec_hits++;
		g.evals++;
		*handled = true;
		return BN_OK;
	}
	// ---- is this a request for the group path at all?  One claim with nothing waiting is the single-claim machinery's.
	// (one claim asked for by more than one pair of sums -- duplicate compositions -- is not: its look-ahead pairs nothing)
	if (rq.k < 2 && rq.terms.size() <= 2 && g.folds.empty() && !g.on) return BN_OK;
	int rc = legacy_to_group(ctx);
	if (rc) return rc;
	if (!ctx->pend_copies.empty()) {
		rc = flush_copies(ctx); // (independent of every deferred fold, checked when they were deferred: any order)
		if (rc) return rc;
	}
	// ---- the request's arrays against the deferred folds
	fold_index fidx;
	fold_ref ref[kMaxArrays];
	for (uint32_t i = 0; i < rq.m; i++) {
		ref[i] = fidx.output(ctx, rq.lo[i], rq.hi[i], rq.row_len);
		if (ref[i].f < 0 && (!group_independent(ctx, rq.lo[i], rq.row_len, false) || !group_independent(ctx, rq.hi[i], rq.row_len, false))) {
			// reads what a deferred fold touches, but not as that fold's output: those folds run first
			rc = group_flush_touching(ctx, rq.lo[i], rq.row_len, false, /*write=*/false);
			if (!rc) rc = group_flush_touching(ctx, rq.hi[i], rq.row_len, false, /*write=*/false);
			if (rc) return rc;
			fidx.reset(); // (indices moved)
			for (uint32_t q = 0; q <= i; q++) ref[q] = fidx.output(ctx, rq.lo[q], rq.hi[q], rq.row_len);
		}
	}
	// a request that reads what a hosted prover's pending write-back covers (but is not that prover's expected call, answered above)
	for (uint32_t i = 0; i < rq.m; i++) {
		rc = unhost_touching(ctx, rq.lo[i], rq.row_len, false);
		if (!rc) rc = unhost_touching(ctx, rq.hi[i], rq.row_len, false);
		if (rc) return rc;
	}
	rc = group_res_alloc(ctx);
	if (rc) return rc;
	pc.lap(bn_ctx::group_state::P_MATCH);
	bn::group_tables *const h_tb = (bn::group_tables *)g.h_tables;
	const bn::group_tables *const d_tb = (const bn::group_tables *)g.d_tables;
	// ---- small enough to finish on the host?  (all arrays contiguous, their deferred folds -- if any -- with one challenge)
	// (the host's rounds cost products in proportion to (claims + arrays / 2) x elements: eight arrays under four claims of 4096
	// elements each is the measured break-even, bn::kGroupTailWorkElems)
	if (g.ht_max && 2 * rq.row_len <= g.ht_max && (uint64_t)rq.m * 2 * rq.row_len <= bn::kGroupTailMaxElems &&
	    ((uint64_t)rq.k + (rq.m + 1) / 2) * 2 * rq.row_len <= g.ht_work && stage_alloc(ctx) == BN_OK) {
		bool ok = true, any_fold = false;
		f128 z{0, 0};
		for (uint32_t i = 0; i < rq.m && ok; i++) {
			ok = rq.hi[i] == (const void *)at(rq.lo[i], rq.row_len);
			if (ok && ref[i].f >= 0) {
				const f128 zi = g.folds[ref[i].f].z;
				if (any_fold && !(zi == z)) ok = false;
				z = zi;
				any_fold = true;
			}
		}
		// (the session first: finding room for it may write another hosted prover's copies back, and that can fail -- before anything
		// of this hand-over has been enqueued, so that a failure leaves every deferred fold exactly as it was)
		bn_ctx::group_session *hs = ok ? session_for(ctx, rq, ref, &rc) : nullptr;
		if (ok && !hs) return rc;
		if (ok) {
			bn::group_mirror_args ma{};
			ma.count = rq.m;
			ma.n = (uint32_t)(2 * rq.row_len);
			ma.z = z;
			for (uint32_t i = 0; i < rq.m; i++) {
				if (ref[i].f >= 0) {
					const auto &f = g.folds[ref[i].f];
					h_tb->mirror.src0[i] = f.src0[ref[i].j];
					h_tb->mirror.x1[i] = f.x1[ref[i].j];
					h_tb->mirror.out[i] = f.x0[ref[i].j];
				} else {
					h_tb->mirror.src0[i] = rq.lo[i];
					h_tb->mirror.x1[i] = nullptr;
					h_tb->mirror.out[i] = nullptr;
				}
			}
			ma.ptrs = &d_tb->mirror;
			ma.staging = (f128 *)g.d_stage;
			ma.phi_tab = (const uint4 *)ctx->d_phi;
			ma.tag_acc = ctx->d_ht_tag;
			ma.counter = ctx->d_ticket;
			ma.mail = ctx->d_mail;
			ctx->mirror.valid = false;
			ma.seq = ++ctx->mail_seq;
			{
				prof_scope ps(ctx, BN_PROF_FOLD_EVAL8);
				const hipError_t e = bn::launch_group_mirror(ctx->stream, ma);
				if (e != hipSuccess) {
					--ctx->mail_seq; // (nothing was enqueued, nothing is retired)
					return bn::hip_fail(e, "launch_group_mirror");
				}
			}
			// retire the folds the hand-over performed; arrays of those batches that the request does not name are folded plainly
			std::vector<std::vector<char>> done(g.folds.size());
			for (size_t f = 0; f < g.folds.size(); f++) done[f].assign(g.folds[f].count, 0);
			for (uint32_t i = 0; i < rq.m; i++)
				if (ref[i].f >= 0) done[ref[i].f][ref[i].j] = 1;
			for (size_t f = g.folds.size(); f-- > 0;) {
				bool any = false;
				for (uint32_t j = 0; j < g.folds[f].count; j++) any = any || done[f][j];
				if (!any) continue;
				bn_ctx::group_fold rest;
				rest.n = g.folds[f].n;
				rest.z = g.folds[f].z;
				for (uint32_t j = 0; j < g.folds[f].count; j++)
					if (!done[f][j]) rest.push(g.folds[f].x0[j], g.folds[f].x1[j], g.folds[f].src0[j]);
				g.folds.erase(g.folds.begin() + (long)f);
				if (rest.count) {
					rc = launch_fold(ctx, rest);
					ctx->grp.flushed_folds--;
					if (rc) return rc;
				}
			}
			g.on = true;
			rc = wait_mail(ctx, ma.seq);
			if (rc) return rc;
			pc.lap(bn_ctx::group_state::P_HOST_WAIT);
			// the staging validates itself (kernels_group.hip k_group_mirror): accepted only when the tag computed from what is read is
			// the tag the kernel published
			record(ctx, *hs, rq);
			hs->hy.resize(rq.m); // (the copies keep their capacity from prove to prove)
			for (uint32_t j = 0; j < rq.m; j++) hs->hy[j].resize(2 * (size_t)ma.n);
			const uint64_t *src = (const uint64_t *)g.h_stage;
			bool valid = false;
			for (int tries = 0; tries < 4096 && !valid; tries++) {
				if (tries == 2048) BN_HIP(hipStreamSynchronize(ctx->stream));
				const uint64_t want = __atomic_load_n(&ctx->h_mail[66].lo, __ATOMIC_ACQUIRE);
				uint64_t t = ma.seq, idx = 0;
				for (uint32_t j = 0; j < rq.m; j++) {
					uint64_t *dst = hs->hy[j].data();
					const uint64_t *sj = src + 2 * (uint64_t)j * ma.n;
					for (uint64_t i = 0; i < ma.n; i++, idx++) {
						const uint64_t lo = __atomic_load_n(&sj[2 * i], __ATOMIC_RELAXED), hi = __atomic_load_n(&sj[2 * i + 1], __ATOMIC_RELAXED);
						dst[2 * i] = lo;
						dst[2 * i + 1] = hi;
						const unsigned r1 = (unsigned)(idx & 63), r2 = (unsigned)((idx * 7 + 17) & 63);
						t ^= ((lo << r1) | (lo >> ((64 - r1) & 63))) ^ ((hi << r2) | (hi >> ((64 - r2) & 63))) ^ (idx + 1) * 0x9E3779B97F4A7C15ull;
					}
				}
				valid = t == want;
			}
			if (!valid) return bn::fail(BN_ERR_DEVICE, "device error: a hosted prover's staging never became consistent");
			pc.lap(bn_ctx::group_state::P_HOST_COPY);
			hs->hosted = true;
			hs->h_len = ma.n;
			hs->h_levels = 0;
			hs->h_n0 = 0;
			hs->h_lo.assign(rq.lo, rq.lo + rq.m);
			hs->h_hi.assign(rq.hi, rq.hi + rq.m);
			hs->h_out.assign(rq.m, nullptr);
			g.hosted_started++;
			host_answer(ctx, *hs, rq, ret_values, n_ret, h_out);
			*handled = true;
			pc.lap(bn_ctx::group_state::P_HOSTED);
			return BN_OK;
		}
	}
	bn_ctx::group_session *self = session_for(ctx, rq, ref, &rc);
	if (!self) return rc;
	// consumed[f][j]: array j of deferred fold f is brought up to date by this launch (a job of it, or a plain launch in front)
	std::vector<std::vector<char>> consumed(g.folds.size());
	for (size_t f = 0; f < g.folds.size(); f++) consumed[f].assign(g.folds[f].count, 0);
	for (uint32_t i = 0; i < rq.m; i++)
		if (ref[i].f >= 0) consumed[ref[i].f][ref[i].j] = 1;
	// ---- the other provers whose fold is waiting and whose claims are known: their next evaluation rides along
	struct rider {
		bn_ctx::group_session *s;
		uint32_t slot0;
		std::vector<fold_ref> ref;
		std::vector<const void *> lo, hi;
		uint64_t row_len;
	};
	std::vector<rider> riders;
	uint32_t n_slots = 2 * rq.k, n_claims = rq.k;
	if (g.speculate) {
		for (auto &s : g.sessions) {
			if (&s == self || s.hosted || s.k == 0 || s.row_len < 2) continue;
			if (n_claims + s.k > (uint32_t)bn::kGroupMaxJobs || n_slots + 2 * s.k > (uint32_t)bn::kGroupMaxSlots) continue;
			rider r{};
			r.s = &s;
			r.row_len = s.row_len / 2;
			r.ref.resize(s.m);
			r.lo.resize(s.m);
			r.hi.resize(s.m);
			bool ok = true;
			fidx.hint = fold_ref{};
			for (uint32_t i = 0; i < s.m && ok; i++) {
				r.ref[i] = fidx.input(ctx, s.lo[i], s.hi[i], s.row_len);
				ok = r.ref[i].f >= 0 && !consumed[r.ref[i].f][r.ref[i].j];
				if (ok) {
					r.lo[i] = g.folds[r.ref[i].f].x0[r.ref[i].j];
					r.hi[i] = at(r.lo[i], r.row_len);
				}
			}
			if (!ok) continue;
			r.slot0 = n_slots;
			n_slots += 2 * s.k;
			n_claims += s.k;
			for (uint32_t i = 0; i < s.m; i++) consumed[r.ref[i].f][r.ref[i].j] = 1;
			riders.push_back(std::move(r));
		}
	}
	// arrays of the participating provers' batches that appear in no request (an unconstrained column) are folded with them, so
	// that the batch can be retired
	std::vector<std::pair<int, int>> loose;
	for (size_t f = 0; f < g.folds.size(); f++) {
		bool any = false;
		for (uint32_t j = 0; j < g.folds[f].count; j++) any = any || consumed[f][j];
		if (!any) continue;
		for (uint32_t j = 0; j < g.folds[f].count; j++)
			if (!consumed[f][j]) {
				consumed[f][j] = 1;
				loose.push_back({(int)f, (int)j});
			}
	}
	// ---- the jobs.  Chained planning (large arrays): every waiting fold is a job of the launch; otherwise, and where that does not
	// fit (more than kGroupMaxJobs jobs): plain fold launches in front for everything a fused job does not fold
	std::vector<bn::group_job> jobs;
	uint64_t n_fused = 0, n_fold_only = 0, n_chains = 0;
	std::vector<std::vector<char>> ran(g.folds.size()); // folded by a plain launch that has been enqueued
	for (size_t f = 0; f < g.folds.size(); f++) ran[f].assign(g.folds[f].count, 0);
	std::vector<std::vector<char>> plain(g.folds.size()); // folded by a plain launch in front of the group launch
	// Chains pay where the arrays are large -- the shared arrays' folds cost no pass of their own -- and lose where a launch is
	// latency-bound: the jobs of a chain run one after the other on one set of workgroups (measured, 2 x 2 claims: 474 against
	// 537 us at 2^24 elements per array, 152 against 143 at 2^22, 39 against 17 at 2^14; profiles/r05).  Per prover by size.
	const size_t cap = (size_t)bn::kGroupMaxJobs; // (more jobs than workgroups: the launcher packs them, kernels_group.hip)
	auto plan_all = [&](bool allow_chains) {
		jobs.clear();
		n_fused = n_fold_only = n_chains = 0;
		for (size_t f = 0; f < g.folds.size(); f++) plain[f].assign(g.folds[f].count, 0);
		auto one = [&](uint32_t m, uint32_t k, const uint16_t *pa, const uint16_t *pb, const void *const *lo, const void *const *hi, uint64_t row_len, const fold_ref *rf,
		               uint32_t slot0) {
			if (jobs.size() > cap) return false;
			if (allow_chains && row_len >= g.chain_min_rows)
				return plan_chained(ctx, m, k, pa, pb, lo, hi, row_len, rf, slot0, cap - jobs.size(), jobs, n_fused, n_fold_only, n_chains);
			std::vector<char> pf(m, 0);
			plan_prefold(ctx, m, k, pa, pb, lo, hi, row_len, rf, slot0, jobs, pf, n_fused);
			for (uint32_t i = 0; i < m; i++)
				if (pf[i]) plain[rf[i].f][rf[i].j] = 1;
			return jobs.size() <= cap;
		};
		if (!one(rq.m, rq.k, rq.pa, rq.pb, rq.lo, rq.hi, rq.row_len, ref, 0)) return false;
		for (const auto &r : riders)
			if (!one(r.s->m, r.s->k, r.s->pa.data(), r.s->pb.data(), r.lo.data(), r.hi.data(), r.row_len, r.ref.data(), r.slot0)) return false;
		for (size_t q = 0; q < loose.size();) {
			const auto &f = g.folds[loose[q].first];
			if (!allow_chains || f.n / 2 < g.chain_min_rows || f.n < 2 || (f.n & 1)) {
				plain[loose[q].first][loose[q].second] = 1;
				q++;
				continue;
			}
			bn::group_job j{};
			j.kind = 3;
			j.n = f.n / 2;
			j.z = f.z;
			const bool two = q + 1 < loose.size() && loose[q + 1].first == loose[q].first;
			for (int sd = 0; sd < (two ? 2 : 1); sd++) {
				const int jj = loose[q + sd].second;
				j.x0[sd] = f.src0[jj];
				j.x1[sd] = f.x1[jj];
				j.out[sd] = f.x0[jj];
			}
			jobs.push_back(j);
			n_fold_only++;
			q += two ? 2 : 1;
		}
		return jobs.size() <= cap;
	};
	if (!plan_all(true) && !plan_all(false)) return BN_OK; // (more jobs than a launch's table holds; nothing has been enqueued: the eager kernels answer)
	{
		std::vector<bn_ctx::group_fold> parts;
		std::vector<size_t> part_of;
		for (size_t f = 0; f < g.folds.size(); f++) {
			bn_ctx::group_fold part;
			part.n = g.folds[f].n;
			part.z = g.folds[f].z;
			for (uint32_t j = 0; j < g.folds[f].count; j++)
				if (plain[f][j]) part.push(g.folds[f].x0[j], g.folds[f].x1[j], g.folds[f].src0[j]);
			if (part.count) {
				parts.push_back(std::move(part));
				part_of.push_back(f);
			}
		}
		// the provers of a batch round fold by ONE challenge (front_loaded.rs:122-155): their plain folds -- arrays of different
		// lengths -- are one launch while the whole is launch-bound
		bool merged = parts.size() >= 2;
		uint64_t arrays = 0, elems = 0;
		for (const auto &pt : parts) {
			merged = merged && pt.z == parts[0].z;
			arrays += pt.count;
			elems += (uint64_t)pt.count * pt.n;
		}
		merged = merged && arrays <= (uint64_t)bn::kFoldBatchMax && elems < ((uint64_t)1 << 25);
		if (merged) {
			bn::fold_batch fb{};
			bn::fold_lengths fl{};
			uint32_t c = 0;
			for (const auto &pt : parts)
				for (uint32_t i = 0; i < pt.count; i++, c++) {
					fb.x0[c] = pt.x0[i];
					fb.x1[c] = pt.x1[i];
					fb.src0[c] = pt.src0[i] != pt.x0[i] ? pt.src0[i] : nullptr;
					fl.n[c] = pt.n;
				}
			{
				prof_scope ps(ctx, BN_PROF_FOLD);
				BN_HIP(bn::launch_extrapolate_line_ragged(ctx->stream, ctx->n_cu, fb, fl, c, parts[0].z));
			}
			g.prefolds++;
		} else {
			for (const auto &pt : parts) {
				rc = launch_fold(ctx, pt);
				ctx->grp.flushed_folds--; // (not a flush: part of the round)
				if (rc) return rc;
				g.prefolds++;
			}
		}
		for (size_t q = 0; q < parts.size(); q++)
			for (uint32_t j = 0; j < g.folds[part_of[q]].count; j++)
				if (plain[part_of[q]][j]) ran[part_of[q]][j] = 1;
	}
	for (const auto &r : riders) g.spec_jobs += r.s->k;
	pc.lap(bn_ctx::group_state::P_PLAN);
	// ---- the launch
	ctx->mirror.valid = false;
	const uint64_t seq = ++ctx->mail_seq;
	{
		prof_scope ps(ctx, (n_fused || n_fold_only) ? BN_PROF_FOLD_EVAL_MFMA : BN_PROF_ROUND_EVAL_MFMA); // (a launch without a fold: round 0)
		const hipError_t e = bn::launch_group(ctx->stream, ctx->n_cu, jobs.data(), (uint32_t)jobs.size(), n_slots, g.d_S, g.d_gmail, ctx->d_mail, ctx->d_ticket, seq,
		                                      h_tb->jobs, d_tb->jobs);
		if (e != hipSuccess) {
			// nothing was enqueued by the failed launch: the folds its jobs would have performed are still deferred; those a plain
			// launch in front has performed (the fallback) are retired.  Then everything runs eagerly.
			--ctx->mail_seq;
			(void)hipGetLastError();
			for (size_t f = g.folds.size(); f-- > 0;) {
				bn_ctx::group_fold rest;
				rest.n = g.folds[f].n;
				rest.z = g.folds[f].z;
				for (uint32_t j = 0; j < g.folds[f].count; j++)
					if (!ran[f][j]) rest.push(g.folds[f].x0[j], g.folds[f].x1[j], g.folds[f].src0[j]);
				if (rest.count)
					g.folds[f] = std::move(rest);
				else
					g.folds.erase(g.folds.begin() + (long)f);
			}
			rc = group_flush(ctx);
			if (rc) return rc;
			if (e == hipErrorNotSupported) return BN_OK; // (the eager kernels answer from up-to-date memory)
			return bn::hip_fail(e, "launch_group");
		}
	}
	g.launches++;
	g.jobs_fused += n_fused;
	g.jobs_fold += n_fold_only;
	g.chain_count += n_chains;
	g.jobs_eval += jobs.size() - n_fused - n_fold_only;
	// retire the consumed arrays
	for (size_t f = g.folds.size(); f-- > 0;) {
		bn_ctx::group_fold rest;
		rest.n = g.folds[f].n;
		rest.z = g.folds[f].z;
		for (uint32_t j = 0; j < g.folds[f].count; j++)
			if (!consumed[f][j]) rest.push(g.folds[f].x0[j], g.folds[f].x1[j], g.folds[f].src0[j]);
		if (rest.count)
			g.folds[f] = std::move(rest);
		else
			g.folds.erase(g.folds.begin() + (long)f);
	}
	g.on = true;
	pc.lap(bn_ctx::group_state::P_LAUNCH);
	rc = wait_mail(ctx, seq);
	if (rc) return rc;
	pc.lap(bn_ctx::group_state::P_WAIT);
	std::vector<f128> raw(n_slots);
	for (uint32_t i = 0; i < n_slots; i++) {
		raw[i].lo = __atomic_load_n(&g.h_gmail[i].lo, __ATOMIC_RELAXED);
		raw[i].hi = __atomic_load_n(&g.h_gmail[i].hi, __ATOMIC_RELAXED);
	}
	answer(rq, raw.data(), ret_values, n_ret, h_out);
	record(ctx, *self, rq);
	for (const auto &r : riders) {
		bn_ctx::group_session &s = *r.s;
		s.pre_valid = true;
		s.pre_row_len = r.row_len;
		s.pre_lo = r.lo;
		s.pre_hi = r.hi;
		s.pre_hull_b = s.pre_hull_e = nullptr;
		for (uint32_t i = 0; i < s.m; i++)
			for (const void *q : {r.lo[i], r.hi[i]}) {
				const char *b0 = (const char *)q, *e0 = b0 + r.row_len * sizeof(f128);
				if (!s.pre_hull_b || b0 < s.pre_hull_b) s.pre_hull_b = b0;
				if (!s.pre_hull_e || e0 > s.pre_hull_e) s.pre_hull_e = e0;
			}
		s.pre_raw.assign(raw.begin() + r.slot0, raw.begin() + r.slot0 + 2 * s.k);
	}
	g.evals++;
	*handled = true;
	pc.lap(bn_ctx::group_state::P_ANSWER);
	return BN_OK;
}

} // namespace bnabi
