// binius_amd/csrc/abi_hal.cpp -- extern "C" entry points of the OLD hardware abstraction layer
// (binius_hal::ComputationBackend, crates/hal/src/backend.rs:35-84) for device-resident multilinears:
// bn_hal_round_evals = sumcheck_compute_round_evals (crates/hal/src/sumcheck_round_calculation.rs:45-330),
// bn_hal_fold_multilinear = one multilinear of sumcheck_fold_multilinears (crates/hal/src/sumcheck_folding.rs:16-237).
//
// Routing: compositions that are sums of monomials of at most three FULL multilinears (two next to an equality-indicator
// table), evaluated at X = 1 and X = infinity in High-to-Low order -- the bivariate products of the v2 provers, the
// zerocheck-style constraints a * b + c -- are sums of what the ComputeLayer path evaluates and run on its kernels, one
// pass per distinct monomial (two factors: matrix-core Gram kernels from 2^17 points, 9-lane kernels below; more: the
// bit-sliced product-sum kernel).  Everything else (more evaluation points, Low-to-High order, truncated multilinears,
// larger compositions) runs the general kernels of kernels_hal.hip.  Transparent multilinears are first partially
// evaluated at the tensor query into context scratch (evaluate_partial_low / _high = the fold_right / fold_left
// kernels): the reference does the same thing subcube by subcube to save memory
// (sumcheck_round_calculation.rs:404-418, 496-518).
#include <algorithm>
#include <chrono>
#include <memory>
#include <map>

#include "abi_common.hpp"
#include "hostmul.hpp"

namespace {

// An ArithCircuit as a sum of monomials  coeff * prod(vars)  over GF(2^128) (variables may repeat: a^2 * b is the
// multiset {a, a, b}).  Empty result = too large for the routed path (more than kMaxTerms monomials / degree > 3 / a
// power above 3): the caller falls back to the interpreter kernel.
typedef bn_expr::monomial monomial;
constexpr size_t kMaxTerms = 12;
int hal_const_tables(bn_ctx *ctx, uint64_t half);
// (flat: a polynomial is a short vector of terms kept sorted by their variables -- a constraint set compiles two hundred of these
// per prove, and ordered maps of vectors cost a prove's zerocheck milliseconds of allocations)
bool expand_poly(const bn_expr *e, std::vector<monomial> &out)
{
	struct term {
		uint32_t n = 0, v[3] = {0, 0, 0};
		f128 c{0, 0};
	};
	typedef std::vector<term> poly;
	auto less = [](const term &x, const term &y) {
		if (x.n != y.n) return x.n < y.n;
		for (uint32_t i = 0; i < x.n; i++)
			if (x.v[i] != y.v[i]) return x.v[i] < y.v[i];
		return false;
	};
	auto same = [](const term &x, const term &y) { return x.n == y.n && (x.n < 1 || x.v[0] == y.v[0]) && (x.n < 2 || x.v[1] == y.v[1]) && (x.n < 3 || x.v[2] == y.v[2]); };
	auto add_term = [&](poly &p, const term &t) {
		if (t.c == bn::f128_zero()) return;
		for (size_t i = 0; i < p.size(); i++)
			if (same(p[i], t)) {
				p[i].c ^= t.c;
				if (p[i].c == bn::f128_zero()) p.erase(p.begin() + (long)i);
				return;
			}
		p.push_back(t);
	};
	auto mul = [&](const poly &x, const poly &y, poly &r) -> bool {
		for (const term &tx : x)
			for (const term &ty : y) {
				if (tx.n + ty.n > 3) return false;
				term t;
				t.n = tx.n + ty.n;
				for (uint32_t i = 0; i < tx.n; i++) t.v[i] = tx.v[i];
				for (uint32_t i = 0; i < ty.n; i++) t.v[tx.n + i] = ty.v[i];
				std::sort(t.v, t.v + t.n);
				t.c = (tx.c == bn::f128_one()) ? ty.c : (ty.c == bn::f128_one()) ? tx.c : bn::mul_host(tx.c, ty.c);
				add_term(r, t);
			}
		return r.size() <= 4 * kMaxTerms;
	};
	std::vector<poly> val(e->steps.size());
	for (size_t i = 0; i < e->steps.size(); i++) {
		const bn_step &st = e->steps[i];
		poly &r = val[i];
		switch (st.kind) {
		case BN_STEP_VAR: {
			term t;
			t.n = 1;
			t.v[0] = st.a;
			t.c = bn::f128_one();
			r.push_back(t);
			break;
		}
		case BN_STEP_CONST: {
			term t;
			t.c = f128{st.cst.lo, st.cst.hi};
			add_term(r, t);
			break;
		}
		case BN_STEP_ADD:
			if (st.a >= i || st.b >= i) return false;
			r = val[st.a];
			for (const term &t : val[st.b]) add_term(r, t);
			break;
		case BN_STEP_MUL:
			if (st.a >= i || st.b >= i) return false;
			if (!mul(val[st.a], val[st.b], r)) return false;
			break;
		case BN_STEP_POW: {
			if (st.a >= i || st.b > 3) return false;
			poly acc;
			term one;
			one.c = bn::f128_one();
			acc.push_back(one);
			for (uint64_t k = 0; k < st.b; k++) {
				poly nxt;
				if (!mul(acc, val[st.a], nxt)) return false;
				acc.swap(nxt);
			}
			r = acc;
			break;
		}
		default: return false;
		}
	}
	out.clear();
	if (e->steps.empty()) return true;
	poly fin = val.back();
	std::sort(fin.begin(), fin.end(), less); // (the order the callers' job lists were built in: shorter monomials first, then by variable)
	for (const term &t : fin) out.push_back(monomial{std::vector<uint32_t>(t.v, t.v + t.n), t.c});
	return out.size() <= kMaxTerms;
}

// Transparent multilinear -> 2^n_vars large-field values at `dst` under the query
int materialise(bn_ctx *ctx, uint32_t order, uint32_t n_vars, const bn_hal_multilinear &ml, const void *d_query, uint32_t query_vars, void *d_one,
                void *dst)
{
	BN_REQUIRE(valid_tower_level(ml.tower_level), "unsupported value of tower_level");
	BN_REQUIRE(ml.n_vars_ml == n_vars + query_vars, "transparent multilinear: n_vars_ml must equal n_vars + query variables");
	BN_REQUIRE(ml.len << (7 - ml.tower_level) == (uint64_t)1 << ml.n_vars_ml, "transparent multilinear: packed length does not match n_vars_ml");
	BN_REQUIRE(query_vars == 0 || d_query, "tensor query missing while a multilinear is still transparent");
	const void *q = query_vars ? d_query : d_one;
	const uint64_t q_len = (uint64_t)1 << query_vars, out_len = (uint64_t)1 << n_vars;
	if (order == BN_ORDER_LOW_TO_HIGH)
		BN_HIP(bn::launch_fold_right(ctx->stream, ctx->n_cu, ml.d_evals, ml.tower_level, q, q_len, dst, out_len));
	else
		BN_HIP(bn::launch_fold_left(ctx->stream, ctx->n_cu, ml.d_evals, ml.tower_level, q, q_len, dst, out_len));
	return BN_OK;
}

// the expansion of a compiled circuit, computed once per bn_expr (nullptr: not a polynomial the routed paths take)
const std::vector<monomial> *poly_of(const bn_expr *e)
{
	if (e->poly_state == 0) e->poly_state = expand_poly(e, e->poly) ? 1 : 2;
	return e->poly_state == 1 ? &e->poly : nullptr;
}

// ---- a constraint set's zerocheck in TWO launches per round --------------------------------------------------------------------
// The reference hands sumcheck_compute_round_evals one equality-indicator evaluator per constraint over every column of the table
// (core/src/constraint_system/prove.rs:431-505; keccak: 100 constraints of degree 2 over 204 columns), evaluation points 1 and
// infinity, High-to-Low, every multilinear Folded and full.  With every composition a sum of monomials of at most two columns,
//     S_e(1) = sum_t coeff_t <eq, prod(vars_t)(upper halves)>,   S_e(inf) = sum_t' coeff_t' <eq, prod(vars_t')(lower + upper halves)>
// and every DISTINCT monomial of the whole set is a job of ONE launch of the claim groups' kernel (kernels_group.hip):
//     {u, w}   an evaluate job (kind 1: both sums at once) over (E_lo | E_hi) and (w_lo | w_hi), with E = eq (.) u on both halves of u
//              -- eq (.) (u_lo + u_hi) = E_lo + E_hi, so its sum at infinity is the three-factor sum the evaluator asks for;
//     {v}, {}  plain inner products with the indicator, two to a job (kind 2): <v_hi, eq> for the sum at 1 -- and <v_lo, eq> beside
//              it only where some composition's form at infinity holds the monomial (a leading form of degree 2 holds none).  The scaled
// columns E are one launch of the batched element-wise product (kernels_mul9.hip k_mul9_jobs) in front; which column of a product
// is scaled is a greedy vertex cover of the products' graph (keccak's chi rows are five-cycles: three scaled columns per row
// instead of five).  The coefficients are applied to the 16-byte sums on the host.  The plan (monomials, cover, terms per
// evaluator) is kept while the same compiled compositions come back.
struct eq_set_plan {
	std::vector<const bn_expr *> key; // (composition, composition_at_infinity) per evaluator
	uint64_t epoch = 0;               // bn::expr_epoch() when the plan was made: no expression has been freed since
	uint32_t n_mls = 0;
	struct mono {
		int u = -1, w = -1; // {}: -1, -1; {v}: v, -1; {u, w}: u = the scaled column
		bool at_inf = false; // some evaluator's form at infinity holds it
		uint32_t s1 = 0, s_lo = 0; // where its sums come back: products: s1 = at 1, s1 + 1 = at infinity; others: s1 = <hi, eq>, s_lo = <lo, eq> (at_inf only)
	};
	struct row_product { // a kind-2 entry: <column half, eq>
		int v;           // the column (-1: the all-ones row)
		bool upper;
		uint32_t slot;
	};
	std::vector<uint32_t> products; // monos with two columns, in job order
	std::vector<row_product> rows;
	uint32_t n_slots = 0;
	std::vector<mono> monos;
	std::vector<uint32_t> scaled; // columns with an E array, in E order
	std::vector<int> e_of;        // column -> its E index (-1: none)
	struct term {
		uint32_t mono;
		f128 coeff;
	};
	std::vector<std::vector<term>> t1, tinf; // per evaluator
};
constexpr uint32_t kEqSetMaxScaled = 256; // (two element-wise jobs each: the pinned table holds 512)
constexpr uint32_t kEqSetMaxMonos = 4096;

std::shared_ptr<eq_set_plan> make_eq_set_plan(const bn_hal_evaluator *evs, uint32_t n_evs, uint32_t n_mls)
{
	auto pl = std::make_shared<eq_set_plan>();
	pl->n_mls = n_mls;
	pl->epoch = bn::expr_epoch();
	std::map<std::vector<uint32_t>, uint32_t> index;
	pl->t1.resize(n_evs);
	pl->tinf.resize(n_evs);
	for (uint32_t e = 0; e < n_evs; e++) {
		pl->key.push_back(evs[e].composition);
		pl->key.push_back(evs[e].composition_at_infinity);
		for (int which = 0; which < 2; which++) {
			const std::vector<monomial> *p = poly_of(which ? evs[e].composition_at_infinity : evs[e].composition);
			if (!p) return nullptr;
			for (const monomial &t : *p) {
				if (t.vars.size() > 2) return nullptr;
				for (uint32_t v : t.vars)
					if (v >= n_mls) return nullptr;
				auto it = index.find(t.vars);
				if (it == index.end()) {
					if (pl->monos.size() >= kEqSetMaxMonos) return nullptr;
					eq_set_plan::mono m;
					if (t.vars.size() >= 1) m.u = (int)t.vars[0];
					if (t.vars.size() == 2) m.w = (int)t.vars[1];
					it = index.emplace(t.vars, (uint32_t)pl->monos.size()).first;
					pl->monos.push_back(m);
				}
				(which ? pl->tinf[e] : pl->t1[e]).push_back(eq_set_plan::term{it->second, t.coeff});
				if (which) pl->monos[it->second].at_inf = true;
			}
		}
	}
	// which column of every product carries the indicator: the column in the most products not yet covered, again and again
	pl->e_of.assign(n_mls, -1);
	std::vector<char> covered(pl->monos.size(), 0);
	for (;;) {
		std::vector<uint32_t> cnt(n_mls, 0);
		bool any = false;
		for (size_t i = 0; i < pl->monos.size(); i++) {
			const auto &m = pl->monos[i];
			if (m.w < 0 || covered[i]) continue;
			any = true;
			cnt[m.u]++;
			if (m.w != m.u) cnt[m.w]++;
		}
		if (!any) break;
		uint32_t best = 0;
		for (uint32_t v = 1; v < n_mls; v++)
			if (cnt[v] > cnt[best]) best = v;
		if (pl->scaled.size() >= kEqSetMaxScaled) return nullptr;
		pl->e_of[best] = (int)pl->scaled.size();
		pl->scaled.push_back(best);
		for (size_t i = 0; i < pl->monos.size(); i++) {
			auto &m = pl->monos[i];
			if (m.w < 0 || covered[i] || (m.u != (int)best && m.w != (int)best)) continue;
			if (m.w == (int)best) std::swap(m.u, m.w); // u = the scaled one
			covered[i] = 1;
		}
	}
	// the jobs' slots: two per product, one per row product
	for (size_t i = 0; i < pl->monos.size(); i++)
		if (pl->monos[i].w >= 0) {
			pl->monos[i].s1 = pl->n_slots;
			pl->n_slots += 2;
			pl->products.push_back((uint32_t)i);
		}
	for (size_t i = 0; i < pl->monos.size(); i++) {
		auto &m = pl->monos[i];
		if (m.w >= 0) continue;
		m.s1 = pl->n_slots++;
		pl->rows.push_back(eq_set_plan::row_product{m.u, true, m.s1});
		if (m.at_inf && m.u >= 0) { // (the all-ones row is the same at both points)
			m.s_lo = pl->n_slots++;
			pl->rows.push_back(eq_set_plan::row_product{m.u, false, m.s_lo});
		}
	}
	if (pl->n_slots + 1 > (uint32_t)bn::kGroupMaxSlots || pl->products.size() + (pl->rows.size() + 1) / 2 > (size_t)bn::kGroupMaxJobs) return nullptr;
	return pl;
}

constexpr int kEqSetDeclined = -2000; // (internal: not this shape -- the caller goes on with the general code)
int round_evals_eq_set(bn_ctx *ctx, uint32_t n_vars, const bn_hal_multilinear *mls, uint32_t n_mls, const bn_hal_evaluator *evs, uint32_t n_evs, bn_f128 *h_out)
{
	const uint64_t full = (uint64_t)1 << n_vars, half = full >> 1;
	const void *eq = n_evs ? evs[0].d_eq_ind : nullptr;
	if (!eq || !ctx->grp.enabled) return kEqSetDeclined;
	// BN_GROUP_PROF=1 (diagnostic): host microseconds of this path by phase, printed every 64 calls
	static thread_local double prof_us[6] = {0, 0, 0, 0, 0, 0};
	static thread_local uint64_t prof_calls = 0;
	auto t_lap = std::chrono::steady_clock::now();
	auto lap = [&](int ph) {
		if (!ctx->grp.prof) return;
		const auto now = std::chrono::steady_clock::now();
		prof_us[ph] += std::chrono::duration<double, std::micro>(now - t_lap).count();
		t_lap = now;
	};
	for (uint32_t k = 0; k < n_mls; k++)
		if (mls[k].kind != BN_HAL_ML_FOLDED || mls[k].len < full || !mls[k].d_evals) return kEqSetDeclined;
	for (uint32_t e = 0; e < n_evs; e++)
		if (evs[e].d_eq_ind != eq || evs[e].eval_point_start < 1 || evs[e].eval_point_end > 3 || !evs[e].composition || !evs[e].composition_at_infinity)
			return kEqSetDeclined;
	// ---- the plan: the last one, if the same compiled compositions came back
	std::shared_ptr<eq_set_plan> pl = std::static_pointer_cast<eq_set_plan>(ctx->hal_set_plan);
	bool same = pl && pl->epoch == bn::expr_epoch() && pl->n_mls == n_mls && pl->key.size() == 2 * (size_t)n_evs;
	for (uint32_t e = 0; e < n_evs && same; e++) same = pl->key[2 * e] == evs[e].composition && pl->key[2 * e + 1] == evs[e].composition_at_infinity;
	if (!same) {
		pl = make_eq_set_plan(evs, n_evs, n_mls);
		ctx->hal_set_plan = pl; // (nullptr: the next call plans again -- and declines again)
		if (!pl) return kEqSetDeclined;
	}
	lap(0);
	int rc = hal_const_tables(ctx, half);
	if (rc) return rc;
	rc = group_res_alloc(ctx);
	if (rc) return rc;
	const char *ones = (const char *)ctx->hal_const, *zeros = ones + ctx->hal_const_half * sizeof(f128);
	// ---- E = eq (.) u on both halves, for every scaled column: one launch
	char *E = nullptr;
	if (!pl->scaled.empty()) {
		E = (char *)bn::ctx_scratch(ctx, pl->scaled.size() * full * sizeof(f128));
		if (!E) return kEqSetDeclined; // (no room: the general code needs less)
		if (!ctx->h_mul_jobs) {
			if (hipHostMalloc(&ctx->h_mul_jobs, 2 * kEqSetMaxScaled * sizeof(bn::mul9_job), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
				(void)hipGetLastError();
				ctx->h_mul_jobs = nullptr;
				return kEqSetDeclined;
			}
			BN_HIP(hipHostGetDevicePointer(&ctx->d_mul_jobs, ctx->h_mul_jobs, 0));
		}
		bn::mul9_job *mj = (bn::mul9_job *)ctx->h_mul_jobs;
		for (size_t k = 0; k < pl->scaled.size(); k++) {
			const char *u = (const char *)mls[pl->scaled[k]].d_evals;
			char *out = E + k * full * sizeof(f128);
			mj[2 * k] = bn::mul9_job{u, eq, out};
			mj[2 * k + 1] = bn::mul9_job{u + half * sizeof(f128), eq, out + half * sizeof(f128)};
		}
		prof_scope ps(ctx, BN_PROF_OTHER);
		BN_HIP(bn::launch_mul9_jobs(ctx->stream, ctx->n_cu, (const bn::mul9_job *)ctx->d_mul_jobs, (uint32_t)(2 * pl->scaled.size()), half));
	}
	lap(1);
	// ---- ONE launch: the products as evaluate jobs, the row products two to a job
	std::vector<f128> sums(pl->n_slots);
	bn::group_tables *h_tb = (bn::group_tables *)ctx->grp.h_tables;
	const bn::group_tables *d_tb = (const bn::group_tables *)ctx->grp.d_tables;
	std::vector<bn::group_job> jobs;
	jobs.reserve(pl->products.size() + (pl->rows.size() + 1) / 2);
	for (uint32_t mi : pl->products) {
		const auto &m = pl->monos[mi];
		bn::group_job j{};
		j.kind = 1;
		j.n = half;
		j.slot = m.s1;
		const char *Ek = E + (size_t)pl->e_of[m.u] * full * sizeof(f128);
		const char *w = (const char *)mls[m.w].d_evals;
		j.x0[0] = Ek;
		j.x1[0] = Ek + half * sizeof(f128);
		j.x0[1] = w;
		j.x1[1] = w + half * sizeof(f128);
		jobs.push_back(j);
	}
	(void)zeros;
	auto row_ptr = [&](const eq_set_plan::row_product &r) -> const char * {
		if (r.v < 0) return ones;
		return (const char *)mls[r.v].d_evals + (r.upper ? half * sizeof(f128) : 0);
	};
	for (size_t q = 0; q < pl->rows.size(); q += 2) {
		// (kind 2 writes S[slot] and S[slot + 1]: the two row products of a job have adjacent slots by construction)
		bn::group_job j{};
		j.kind = 2;
		j.n = half;
		j.slot = pl->rows[q].slot;
		j.x0[0] = row_ptr(pl->rows[q]);
		j.x0[1] = eq;
		if (q + 1 < pl->rows.size()) {
			j.x1[0] = row_ptr(pl->rows[q + 1]);
			j.x1[1] = eq;
		}
		jobs.push_back(j);
	}
	lap(2);
	if (!jobs.empty()) {
		ctx->mirror.valid = false;
		const uint64_t seq = ++ctx->mail_seq;
		{
			prof_scope ps(ctx, BN_PROF_ROUND_EVAL_MFMA);
			// (an odd last row product leaves its job's second slot unused: one slot more than the plan counts)
			const hipError_t e = bn::launch_group(ctx->stream, ctx->n_cu, jobs.data(), (uint32_t)jobs.size(), pl->n_slots + 1, ctx->grp.d_S, ctx->grp.d_gmail, ctx->d_mail,
			                                      ctx->d_ticket, seq, h_tb->jobs, d_tb->jobs);
			if (e != hipSuccess) {
				--ctx->mail_seq;
				(void)hipGetLastError();
				if (e == hipErrorNotSupported) return kEqSetDeclined;
				return bn::hip_fail(e, "launch_group (constraint set)");
			}
		}
		lap(3);
		volatile uint64_t *seqw = &ctx->h_mail[64].lo;
		uint64_t spins = 0;
		while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) {
			if (++spins > (1ull << 22)) {
				BN_HIP(hipStreamSynchronize(ctx->stream));
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
				break;
			}
		}
		lap(4);
		for (uint32_t i = 0; i < pl->n_slots; i++) {
			sums[i].lo = __atomic_load_n(&ctx->grp.h_gmail[i].lo, __ATOMIC_RELAXED);
			sums[i].hi = __atomic_load_n(&ctx->grp.h_gmail[i].hi, __ATOMIC_RELAXED);
		}
	}
	// ---- the evaluators' values: coefficient x sum, the sum at 1 or at infinity
	size_t off = 0;
	for (uint32_t e = 0; e < n_evs; e++)
		for (uint32_t p = evs[e].eval_point_start; p < evs[e].eval_point_end; p++, off++) {
			f128 v = bn::f128_zero();
			for (const auto &t : (p == 1 ? pl->t1[e] : pl->tinf[e])) {
				const auto &m = pl->monos[t.mono];
				f128 sv = sums[m.s1];
				if (p == 2) {
					if (m.w >= 0)
						sv = sums[m.s1 + 1];
					else if (m.u >= 0)
						sv = sv ^ sums[m.s_lo]; // <v_lo + v_hi, eq>
				}
				v ^= (t.coeff == bn::f128_one()) ? sv : bn::mul_host(t.coeff, sv);
			}
			h_out[off] = bn_f128{v.lo, v.hi};
		}
	lap(5);
	if (ctx->grp.prof && (++prof_calls & 63) == 0)
		fprintf(stderr, "[bn eq-set prof] %llu calls, host us: plan %.1f, tables + scaling launch %.1f, jobs %.1f, sums launch %.1f, wait %.1f, read + values %.1f\n",
		        (unsigned long long)prof_calls, prof_us[0], prof_us[1], prof_us[2], prof_us[3], prof_us[4], prof_us[5]);
	return BN_OK;
}

// ---- compositions of degree <= 3 at ANY evaluation points: the round polynomial's COEFFICIENTS from bilinear sums ---------------------
// High-to-Low, every multilinear Folded and full: v(X) = v_lo + X dv (dv = v_lo + v_hi), so a monomial's sum over the half cube is a
// polynomial in X whose coefficients are sums of products of HALVES -- no row is ever brought to a domain point z on the device (the
// general code below multiplies every row by z through nibble tables, once per point), and any number of domain points costs the
// same: the 16-byte coefficients meet z on the host.  With [x, y] = sum_i x[i] y[i] over the half cube:
//     {v}        k0 = [v_lo, e]   k1 = [dv, e]                                   (e = the indicator, or the all-ones row)
//     {u, w}     k0 = [u_lo, w_lo]   k2 = [du, dw]   P(1) = [u_hi, w_hi]   k1 = P(1) + k0 + k2
//     {a, b, c}  T0 = a_lo (.) b_lo, T1 = a_hi (.) b_hi, Ti = da (.) db  (element-wise, ONE launch of k_mul9_jobs for every distinct pair
//                of the request: the Karatsuba triple of a (.) b as a polynomial in X)
//                P(1) = [T1, c_hi]   k3 = [Ti, dc]   k0 = [T0, c_lo]   k2 = [Ti, c_hi] + [T0 + T1, dc]   k1 = P(1) + k0 + k2 + k3
// (a (.) b = T0 + X (T0 + T1 + Ti) + X^2 Ti; multiply by c_lo + X dc and collect).  Under an equality indicator the first column of
// every monomial is scaled by it first (both halves: eq (.) du = E_lo + E_hi), as the constraint-set path above does.  Every bracket is
// a job of ONE launch of the claim groups' kernel: [x_hi, y_hi] and [dx, dy] together are an evaluate job (kind 1), a lone bracket half
// a kind-2 job.  So a request is at most three launches (indicator scaling, products, sums) whatever its points and compositions.
// Reference: sumcheck_round_calculation.rs:85-330 evaluates the composition at every point of every vertex pair instead.
constexpr int kCoefDeclined = -2001; // (internal: not this shape)
constexpr uint32_t kCoefMaxMulJobs = 2 * kEqSetMaxScaled; // the pinned element-wise job table
int round_evals_coef(bn_ctx *ctx, uint32_t n_vars, const bn_hal_multilinear *mls, uint32_t n_mls, const bn_hal_evaluator *evs, uint32_t n_evs, const bn_f128 *h_points,
                     uint32_t n_points, bn_f128 *h_out)
{
	if (!ctx->grp.enabled || n_vars < 2 || n_evs == 0) return kCoefDeclined;
	const uint64_t full = (uint64_t)1 << n_vars, half = full >> 1;
	for (uint32_t k = 0; k < n_mls; k++)
		if (mls[k].kind != BN_HAL_ML_FOLDED || mls[k].len < full || !mls[k].d_evals) return kCoefDeclined;
	const void *eq = evs[0].d_eq_ind;
	uint32_t pt_hi = 0, total = 0;
	for (uint32_t e = 0; e < n_evs; e++) {
		if (evs[e].d_eq_ind != eq || !evs[e].composition || !evs[e].composition_at_infinity || evs[e].eval_point_start < 1 || evs[e].eval_point_start > evs[e].eval_point_end)
			return kCoefDeclined;
		if (evs[e].composition->n_vars > n_mls || evs[e].composition_at_infinity->n_vars > n_mls) return kCoefDeclined;
		pt_hi = std::max(pt_hi, evs[e].eval_point_end);
		total += evs[e].eval_point_end - evs[e].eval_point_start;
	}
	if (n_points != (pt_hi > 3 ? pt_hi - 3 : 0) || (n_points && !h_points) || total == 0) return kCoefDeclined; // (the general code reports the error)
	const bool has_z = pt_hi > 3;
	// ---- the distinct monomials of the request
	struct mono {
		uint32_t n = 0, v[3] = {0, 0, 0};
		bool in1 = false, in_inf = false; // some evaluator wants it at 1 / a domain point; at infinity
		int pair = -1;                    // n = 3: its product triple
		uint32_t s_a = 0, s_b = 0, s_c = 0; // first slots of its jobs
	};
	struct term {
		uint32_t mono;
		f128 coeff;
	};
	std::vector<mono> monos;
	std::map<std::vector<uint32_t>, uint32_t> index;
	std::vector<std::vector<term>> t1(n_evs), tinf(n_evs);
	uint32_t deg_max = 0;
	for (uint32_t e = 0; e < n_evs; e++) {
		const bool want1 = (evs[e].eval_point_start <= 1 && evs[e].eval_point_end > 1) || evs[e].eval_point_end > 3, want_inf = evs[e].eval_point_start <= 2 && evs[e].eval_point_end > 2;
		for (int which = 0; which < 2; which++) {
			if (!(which ? want_inf : want1)) continue;
			const std::vector<monomial> *p = poly_of(which ? evs[e].composition_at_infinity : evs[e].composition);
			if (!p) return kCoefDeclined;
			for (const monomial &t : *p) {
				if (t.vars.size() > 3) return kCoefDeclined;
				for (uint32_t v : t.vars)
					if (v >= n_mls) return kCoefDeclined;
				auto it = index.find(t.vars);
				if (it == index.end()) {
					mono m;
					m.n = (uint32_t)t.vars.size();
					for (uint32_t i = 0; i < m.n; i++) m.v[i] = t.vars[i];
					it = index.emplace(t.vars, (uint32_t)monos.size()).first;
					monos.push_back(m);
					deg_max = std::max(deg_max, m.n);
				}
				(which ? tinf[e] : t1[e]).push_back(term{it->second, t.coeff});
				(which ? monos[it->second].in_inf : monos[it->second].in1) = true;
			}
		}
	}
	// (points 1 and infinity of compositions of degree <= 2 are what the routed code and the constraint-set path above are tuned for)
	if (!has_z && deg_max < 3) return kCoefDeclined;
	// ---- scaled columns (under an indicator: the first column of every monomial of two or three), product triples
	std::vector<int> e_of(n_mls, -1);
	std::vector<uint32_t> scaled;
	struct triple {
		uint32_t x, y; // columns; x is the scaled one under an indicator
		bool t0 = false, t1 = false, ti = false;
	};
	std::vector<triple> pairs;
	std::map<std::pair<uint32_t, uint32_t>, int> pair_index;
	for (mono &m : monos) {
		if (m.n < 2) continue;
		if (eq && e_of[m.v[0]] < 0) {
			e_of[m.v[0]] = (int)scaled.size();
			scaled.push_back(m.v[0]);
		}
		if (m.n == 3) {
			auto key = std::make_pair(m.v[0], m.v[1]);
			auto it = pair_index.find(key);
			if (it == pair_index.end()) {
				it = pair_index.emplace(key, (int)pairs.size()).first;
				pairs.push_back(triple{m.v[0], m.v[1]});
			}
			m.pair = it->second;
			triple &tp = pairs[(size_t)m.pair];
			tp.t1 = tp.t1 || m.in1;
			tp.t0 = tp.t0 || (m.in1 && has_z);
			tp.ti = tp.ti || m.in_inf || (m.in1 && has_z);
		}
	}
	if (2 * scaled.size() + 3 * pairs.size() > kCoefMaxMulJobs) return kCoefDeclined;
	// ---- the jobs of the sums' launch and their slots: evaluate jobs (two slots each) first, then the lone brackets two to a job
	struct bracket {
		const void *x, *y;
	};
	std::vector<bn::group_job> jobs;
	std::vector<bracket> rows;
	uint32_t n_eval_jobs = 0, n_rows = 0, n_lone = 0;
	for (const mono &m : monos) n_eval_jobs += m.n == 2 ? 1 : 0; // (a first count: is there a launch for the lone sums to ride on)
	for (const mono &m : monos)
		if (m.n == 3 && (m.in_inf || (m.in1 && has_z))) n_eval_jobs++;
	// the sum of a lone row without an indicator: at streaming speed into its slot (k_xor_sum), which the sums' launch publishes with
	// its own -- from 2^20 points a bracket with the all-ones row (twice the bytes, and Gram products of ones) costs more than a launch
	const bool lone_direct = !eq && half >= ((uint64_t)1 << 20) && n_eval_jobs > 0;
	n_eval_jobs = 0;
	for (const mono &m : monos) {
		if (m.n == 0) n_rows += eq ? 1 : 0;
		if (m.n == 1) (lone_direct ? n_lone : n_rows) += 2;
		if (m.n == 2) n_eval_jobs += 1, n_rows += (has_z && m.in1) ? 1 : 0;
		if (m.n == 3) {
			if (m.in1) has_z ? n_eval_jobs++ : n_rows++;
			if (m.in_inf || (m.in1 && has_z)) n_eval_jobs++;
			if (m.in1 && has_z) n_rows++;
		}
	}
	const uint32_t rows_base = 2 * n_eval_jobs, lone_base = rows_base + n_rows + (n_rows & 1), n_slots = lone_base + n_lone;
	if (n_slots + 1 > (uint32_t)bn::kGroupMaxSlots || n_eval_jobs + (n_rows + 1) / 2 > (uint32_t)bn::kGroupMaxJobs) return kCoefDeclined;
	if (n_eval_jobs + n_rows == 0) { // (constants without an indicator: every sum over the half cube is zero)
		for (uint32_t i = 0; i < total; i++) h_out[i] = bn_f128{0, 0};
		return BN_OK;
	}
	int rc = hal_const_tables(ctx, half);
	if (rc) return rc;
	rc = group_res_alloc(ctx);
	if (rc) return rc;
	const char *ones = (const char *)ctx->hal_const, *zeros = ones + ctx->hal_const_half * sizeof(f128);
	const size_t hb = half * sizeof(f128);
	char *scr = nullptr;
	if (!scaled.empty() || !pairs.empty()) {
		scr = (char *)bn::ctx_scratch(ctx, (scaled.size() * 2 + pairs.size() * 3) * hb);
		if (!scr) return kCoefDeclined; // (no room: the general code reports it or needs less)
		if (!ctx->h_mul_jobs) {
			if (hipHostMalloc(&ctx->h_mul_jobs, kCoefMaxMulJobs * sizeof(bn::mul9_job), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
				(void)hipGetLastError();
				ctx->h_mul_jobs = nullptr;
				return kCoefDeclined;
			}
			BN_HIP(hipHostGetDevicePointer(&ctx->d_mul_jobs, ctx->h_mul_jobs, 0));
		}
	}
	char *E = scr, *T = scr + scaled.size() * 2 * hb;
	auto lo_of = [&](uint32_t v, bool maybe_scaled) -> const char * {
		return (maybe_scaled && eq) ? E + (size_t)e_of[v] * 2 * hb : (const char *)mls[v].d_evals;
	};
	// ---- element-wise launches: the indicator's columns, then the product triples (the second reads what the first wrote: its jobs sit
	// behind the first launch's in the pinned table, which the first launch may still be reading)
	bn::mul9_job *mj = (bn::mul9_job *)ctx->h_mul_jobs;
	uint32_t n_mj = 0;
	if (!scaled.empty()) {
		for (size_t k = 0; k < scaled.size(); k++) {
			const char *u = (const char *)mls[scaled[k]].d_evals;
			char *out = E + k * 2 * hb;
			mj[n_mj++] = bn::mul9_job{u, eq, out, nullptr, nullptr};
			mj[n_mj++] = bn::mul9_job{u + hb, eq, out + hb, nullptr, nullptr};
		}
		prof_scope ps(ctx, BN_PROF_OTHER);
		BN_HIP(bn::launch_mul9_jobs(ctx->stream, ctx->n_cu, (const bn::mul9_job *)ctx->d_mul_jobs, n_mj, half));
	}
	if (!pairs.empty()) {
		const uint32_t first = n_mj;
		for (size_t k = 0; k < pairs.size(); k++) {
			const triple &tp = pairs[k];
			const char *x = lo_of(tp.x, true), *y = (const char *)mls[tp.y].d_evals;
			char *out = T + k * 3 * hb;
			if (tp.t0) mj[n_mj++] = bn::mul9_job{x, y, out, nullptr, nullptr};
			if (tp.t1) mj[n_mj++] = bn::mul9_job{x + hb, y + hb, out + hb, nullptr, nullptr};
			if (tp.ti) mj[n_mj++] = bn::mul9_job{x, y, out + 2 * hb, x + hb, y + hb};
		}
		prof_scope ps(ctx, BN_PROF_OTHER);
		BN_HIP(bn::launch_mul9_jobs(ctx->stream, ctx->n_cu, (const bn::mul9_job *)ctx->d_mul_jobs + first, n_mj - first, half));
	}
	// ---- the sums
	uint32_t next_eval = 0, next_lone = 0;
	auto eval_job = [&](const void *x_lo, const void *x_hi, const void *y_lo, const void *y_hi) -> uint32_t {
		bn::group_job j{};
		j.kind = 1;
		j.n = half;
		j.slot = 2 * next_eval++;
		j.x0[0] = x_lo;
		j.x1[0] = x_hi;
		j.x0[1] = y_lo;
		j.x1[1] = y_hi;
		jobs.push_back(j);
		return j.slot;
	};
	auto row = [&](const void *x, const void *y) -> uint32_t {
		rows.push_back(bracket{x, y});
		return rows_base + (uint32_t)rows.size() - 1;
	};
	const void *e_row = eq ? eq : (const void *)ones;
	for (mono &m : monos) {
		if (m.n == 0) {
			if (eq) m.s_a = row(ones, eq);
		} else if (m.n == 1) {
			const char *v = (const char *)mls[m.v[0]].d_evals;
			if (lone_direct) {
				m.s_a = lone_base + next_lone++;
				m.s_b = lone_base + next_lone++;
				prof_scope ps(ctx, BN_PROF_OTHER);
				BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, v + hb, half, ctx->grp.d_S + m.s_a));
				BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, v, half, ctx->grp.d_S + m.s_b));
			} else {
				m.s_a = row(v + hb, e_row);
				m.s_b = row(v, e_row);
			}
		} else if (m.n == 2) {
			const char *u = lo_of(m.v[0], true), *w = (const char *)mls[m.v[1]].d_evals;
			m.s_a = eval_job(u, u + hb, w, w + hb);
			if (has_z && m.in1) m.s_c = row(u, w);
		} else {
			const char *t0 = T + (size_t)m.pair * 3 * hb, *t1p = t0 + hb, *ti = t0 + 2 * hb, *c = (const char *)mls[m.v[2]].d_evals;
			if (m.in1) m.s_a = has_z ? eval_job(t0, t1p, c, c + hb) : row(t1p, c + hb);
			if (m.in_inf || (m.in1 && has_z)) m.s_b = eval_job(zeros, ti, c, c + hb);
			if (m.in1 && has_z) m.s_c = row(t0, c);
		}
	}
	for (size_t q = 0; q < rows.size(); q += 2) {
		bn::group_job j{};
		j.kind = 2;
		j.n = half;
		j.slot = rows_base + (uint32_t)q;
		j.x0[0] = rows[q].x;
		j.x0[1] = rows[q].y;
		if (q + 1 < rows.size()) {
			j.x1[0] = rows[q + 1].x;
			j.x1[1] = rows[q + 1].y;
		}
		jobs.push_back(j);
	}
	std::vector<f128> S(n_slots + 1, bn::f128_zero());
	{
		bn::group_tables *h_tb = (bn::group_tables *)ctx->grp.h_tables;
		const bn::group_tables *d_tb = (const bn::group_tables *)ctx->grp.d_tables;
		ctx->mirror.valid = false;
		const uint64_t seq = ++ctx->mail_seq;
		{
			prof_scope ps(ctx, BN_PROF_ROUND_EVAL_MFMA);
			const hipError_t le = bn::launch_group(ctx->stream, ctx->n_cu, jobs.data(), (uint32_t)jobs.size(), n_slots + 1, ctx->grp.d_S, ctx->grp.d_gmail, ctx->d_mail,
			                                       ctx->d_ticket, seq, h_tb->jobs, d_tb->jobs);
			if (le != hipSuccess) {
				--ctx->mail_seq;
				(void)hipGetLastError();
				if (next_lone) BN_HIP(hipMemsetAsync(ctx->grp.d_S + lone_base, 0, next_lone * sizeof(f128), ctx->stream)); // (the slots are zero between launches)
				if (le == hipErrorNotSupported) return kCoefDeclined; // (the element-wise launches wrote scratch only)
				return bn::hip_fail(le, "launch_group (coefficient form)");
			}
			ctx->grp.launches++;
			ctx->grp.jobs_eval += jobs.size();
		}
		volatile uint64_t *seqw = &ctx->h_mail[64].lo;
		uint64_t spins = 0;
		while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) {
			if (++spins > (1ull << 22)) {
				BN_HIP(hipStreamSynchronize(ctx->stream));
				if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != seq) return bn::fail(BN_ERR_DEVICE, "device error: result mailbox was not published");
				break;
			}
		}
		for (uint32_t i = 0; i < n_slots; i++) {
			S[i].lo = __atomic_load_n(&ctx->grp.h_gmail[i].lo, __ATOMIC_RELAXED);
			S[i].hi = __atomic_load_n(&ctx->grp.h_gmail[i].hi, __ATOMIC_RELAXED);
		}
	}
	// ---- coefficients k0 .. k_deg and P(1) per monomial, then the evaluators' values
	struct coefs {
		f128 k[4], p1;
	};
	std::vector<coefs> cf(monos.size());
	for (size_t i = 0; i < monos.size(); i++) {
		const mono &m = monos[i];
		coefs &c = cf[i];
		for (auto &x : c.k) x = bn::f128_zero();
		c.p1 = bn::f128_zero();
		if (m.n == 0) {
			if (eq) c.k[0] = c.p1 = S[m.s_a];
		} else if (m.n == 1) {
			c.p1 = S[m.s_a];
			c.k[0] = S[m.s_b];
			c.k[1] = S[m.s_a] ^ S[m.s_b];
		} else if (m.n == 2) {
			c.p1 = S[m.s_a];
			c.k[2] = S[m.s_a + 1];
			if (has_z && m.in1) {
				c.k[0] = S[m.s_c];
				c.k[1] = c.p1 ^ c.k[0] ^ c.k[2];
			}
		} else {
			if (m.in1) c.p1 = S[m.s_a];
			if (m.in_inf || (m.in1 && has_z)) c.k[3] = S[m.s_b + 1];
			if (m.in1 && has_z) {
				c.k[0] = S[m.s_c];
				c.k[2] = S[m.s_b] ^ S[m.s_a + 1];
				c.k[1] = c.p1 ^ c.k[0] ^ c.k[2] ^ c.k[3];
			}
		}
	}
	size_t off = 0;
	for (uint32_t e = 0; e < n_evs; e++)
		for (uint32_t p = evs[e].eval_point_start; p < evs[e].eval_point_end; p++, off++) {
			f128 v = bn::f128_zero();
			if (p == 2) {
				for (const term &t : tinf[e]) {
					const f128 sv = cf[t.mono].k[monos[t.mono].n];
					v ^= (t.coeff == bn::f128_one()) ? sv : bn::mul_host(t.coeff, sv);
				}
			} else if (p == 1) {
				for (const term &t : t1[e]) v ^= (t.coeff == bn::f128_one()) ? cf[t.mono].p1 : bn::mul_host(t.coeff, cf[t.mono].p1);
			} else {
				// Horner in z over the coefficient sums of the whole composition (coefficients first: four products per point)
				f128 K[4] = {bn::f128_zero(), bn::f128_zero(), bn::f128_zero(), bn::f128_zero()};
				for (const term &t : t1[e])
					for (uint32_t d = 0; d <= monos[t.mono].n; d++) K[d] ^= (t.coeff == bn::f128_one()) ? cf[t.mono].k[d] : bn::mul_host(t.coeff, cf[t.mono].k[d]);
				const f128 z = to_f(&h_points[p - 3]);
				v = K[3];
				for (int d = 2; d >= 0; d--) v = bn::mul_host(v, z) ^ K[d];
			}
			h_out[off] = bn_f128{v.lo, v.hi};
		}
	return BN_OK;
}

// all-ones | all-zeros tables of at least `half` elements each (ones at hal_const, zeros hal_const_half elements behind it): filled
// once for the largest size asked for so far and kept in the context -- the rounds of a sumcheck ask for halving sizes
int hal_const_tables(bn_ctx *ctx, uint64_t half)
{
	if (ctx->hal_const && ctx->hal_const_half >= half) return BN_OK;
	if (ctx->hal_const) {
		BN_HIP(hipStreamSynchronize(ctx->stream));
		BN_HIP(hipFree(ctx->hal_const));
		ctx->hal_const = nullptr;
		ctx->hal_const_half = 0;
	}
	if (hipMalloc(&ctx->hal_const, 2 * half * sizeof(f128)) != hipSuccess) {
		(void)hipGetLastError();
		ctx->hal_const = nullptr;
		return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (constant tables of the round evaluation)");
	}
	BN_HIP(bn::launch_fill(ctx->stream, ctx->hal_const, half, bn::f128_one()));
	BN_HIP(hipMemsetAsync((char *)ctx->hal_const + half * sizeof(f128), 0, half * sizeof(f128), ctx->stream));
	ctx->hal_const_half = half;
	return BN_OK;
}

// A request wider than one pass of the code below carries (a constraint set's zerocheck: one evaluator per constraint over every
// column of the table, core/src/constraint_system/prove.rs:431-505 -- keccak: a hundred compositions over two hundred
// multilinears, each composition reading four or five of them): the evaluators are dealt out, in order, to parts of at most
// kHalMaxEv evaluators that together read at most kHalMaxMl multilinears and ask for at most 32 values; every part is the same call
// with its compositions' variables renumbered to the multilinears it reads.  The results are those of the whole: an evaluator's
// values depend on nothing but its own compositions and multilinears (sumcheck_round_calculation.rs:85-330).
int round_evals_in_parts(bn_ctx *ctx, uint32_t order, uint32_t n_vars, const void *d_tensor_query, uint32_t query_vars, const bn_hal_multilinear *mls, uint32_t n_mls,
                         const bn_hal_evaluator *evs, uint32_t n_evs, const bn_f128 *h_points, uint32_t n_points, bn_f128 *h_out)
{
	uint32_t pt_hi_all = 0;
	for (uint32_t e = 0; e < n_evs; e++) {
		BN_REQUIRE(evs[e].composition && evs[e].composition_at_infinity, "evaluator without a composition");
		BN_REQUIRE(evs[e].eval_point_start <= evs[e].eval_point_end, "empty evaluation point range");
		BN_REQUIRE(evs[e].composition->n_vars <= n_mls && evs[e].composition_at_infinity->n_vars <= n_mls,
		           "composition uses more variables than there are multilinears");
		if (evs[e].eval_point_end > pt_hi_all) pt_hi_all = evs[e].eval_point_end;
	}
	BN_REQUIRE(n_points == (pt_hi_all > 3 ? pt_hi_all - 3 : 0), "nontrivial evaluation points: incorrect length");
	std::vector<int> local_of(n_mls, -1);
	size_t out_at = 0;
	uint32_t e0 = 0;
	while (e0 < n_evs) {
		std::vector<uint32_t> used;
		uint32_t e1 = e0, pts = 0, pt_hi = 0;
		while (e1 < n_evs && e1 - e0 < (uint32_t)bn::kHalMaxEv) {
			std::vector<uint32_t> fresh;
			for (const bn_expr *c : {(const bn_expr *)evs[e1].composition, (const bn_expr *)evs[e1].composition_at_infinity})
				for (const bn_step &st : c->steps)
					if (st.kind == BN_STEP_VAR && local_of[st.a] < 0 && std::find(fresh.begin(), fresh.end(), st.a) == fresh.end()) fresh.push_back(st.a);
			const uint32_t cnt = evs[e1].eval_point_end - evs[e1].eval_point_start;
			const bool fits = used.size() + fresh.size() <= (size_t)bn::kHalMaxMl && pts + cnt <= 32;
			if (!fits && e1 > e0) break;
			BN_REQUIRE(fits, "one evaluator reads more multilinears or asks for more values than a pass carries");
			for (uint32_t v : fresh) {
				local_of[v] = (int)used.size();
				used.push_back(v);
			}
			pts += cnt;
			if (evs[e1].eval_point_end > pt_hi) pt_hi = evs[e1].eval_point_end;
			e1++;
		}
		std::vector<bn_hal_multilinear> part_mls;
		for (uint32_t v : used) part_mls.push_back(mls[v]);
		std::vector<bn_expr *> owned;
		std::vector<bn_hal_evaluator> part_evs;
		int rc = BN_OK;
		for (uint32_t e = e0; e < e1 && !rc; e++) {
			bn_hal_evaluator pe = evs[e];
			for (int which = 0; which < 2 && !rc; which++) {
				const bn_expr *c = which ? evs[e].composition_at_infinity : evs[e].composition;
				std::vector<bn_step> steps = c->steps;
				for (bn_step &st : steps)
					if (st.kind == BN_STEP_VAR) st.a = (uint32_t)local_of[st.a];
				bn_expr *re = nullptr;
				rc = bn_expr_compile(ctx, steps.data(), steps.size(), &re);
				if (rc) break;
				owned.push_back(re);
				(which ? pe.composition_at_infinity : pe.composition) = re;
			}
			part_evs.push_back(pe);
		}
		std::vector<bn_f128> part_out(pts ? pts : 1);
		if (!rc)
			rc = bn_hal_round_evals(ctx, order, n_vars, d_tensor_query, query_vars, part_mls.data(), (uint32_t)part_mls.size(), part_evs.data(), (uint32_t)part_evs.size(),
			                        h_points, pt_hi > 3 ? pt_hi - 3 : 0, part_out.data());
		for (bn_expr *x : owned) bn_expr_free(x);
		if (rc) return rc;
		for (uint32_t i = 0; i < pts; i++) h_out[out_at + i] = part_out[i];
		out_at += pts;
		for (uint32_t v : used) local_of[v] = -1;
		e0 = e1;
	}
	return BN_OK;
}

} // namespace

extern "C" {

int bn_hal_round_evals(bn_ctx *ctx, uint32_t order, uint32_t n_vars, const void *d_tensor_query, uint32_t query_vars,
                       const bn_hal_multilinear *mls, uint32_t n_mls, const bn_hal_evaluator *evs, uint32_t n_evs,
                       const bn_f128 *h_points, uint32_t n_points, bn_f128 *h_out)
{
	BN_REQUIRE(ctx && h_out && (mls || n_mls == 0) && (evs || n_evs == 0), "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(order == BN_ORDER_LOW_TO_HIGH || order == BN_ORDER_HIGH_TO_LOW, "unknown evaluation order");
	BN_REQUIRE(n_vars > 0 && n_vars < 40, "computing round evaluations requires at least a single variable");
	if (order == BN_ORDER_HIGH_TO_LOW && ctx->hal_coef) {
		// compositions of degree <= 3 over full Folded multilinears at domain points (or of degree 3 at 1 / infinity): the round
		// polynomial's coefficients from bilinear sums, at most three launches whatever the width of the request
		const int rc_c = round_evals_coef(ctx, n_vars, mls, n_mls, evs, n_evs, h_points, n_points, h_out);
		if (rc_c != kCoefDeclined) return rc_c;
	}
	{
		uint32_t pts = 0;
		for (uint32_t e = 0; e < n_evs; e++) pts += evs[e].eval_point_end > evs[e].eval_point_start ? evs[e].eval_point_end - evs[e].eval_point_start : 0;
		const bool wide = n_mls > (uint32_t)bn::kHalMaxMl || n_evs > (uint32_t)bn::kHalMaxEv || pts > 32;
		// (a narrow request under an indicator takes the constraint set's two launches too: the routed code below runs one pass per
		// monomial and reads a lone column's lower half beside an all-zeros table -- measured, (a b + c) eq: n = 13 48 -> 39 us,
		// n = 20 96 -> 77 us, n = 24 0.62 -> 0.60 ms, where the 2^24 products of the scaling are what is left)
		if (wide || (n_evs && evs[0].d_eq_ind)) {
			if (order == BN_ORDER_HIGH_TO_LOW && n_points == 0 && ctx->hal_eq_set) {
				const int rc_s = round_evals_eq_set(ctx, n_vars, mls, n_mls, evs, n_evs, h_out);
				if (rc_s != kEqSetDeclined) return rc_s;
			}
			if (wide) return round_evals_in_parts(ctx, order, n_vars, d_tensor_query, query_vars, mls, n_mls, evs, n_evs, h_points, n_points, h_out);
		}
	}
	uint32_t pt_lo = 0, pt_hi = 0, total = 0;
	for (uint32_t e = 0; e < n_evs; e++) {
		BN_REQUIRE(evs[e].composition && evs[e].composition_at_infinity, "evaluator without a composition");
		BN_REQUIRE(evs[e].eval_point_start <= evs[e].eval_point_end, "empty evaluation point range");
		BN_REQUIRE(evs[e].composition->n_vars <= n_mls && evs[e].composition_at_infinity->n_vars <= n_mls,
		           "composition uses more variables than there are multilinears");
		BN_REQUIRE(evs[e].composition->steps.size() <= 64 && evs[e].composition_at_infinity->steps.size() <= 64, "composition too large");
		if (e == 0 || evs[e].eval_point_start < pt_lo) pt_lo = evs[e].eval_point_start;
		if (e == 0 || evs[e].eval_point_end > pt_hi) pt_hi = evs[e].eval_point_end;
		total += evs[e].eval_point_end - evs[e].eval_point_start;
	}
	// Error::IncorrectNontrivialEvalPointsLength (sumcheck_round_calculation.rs:121-125)
	BN_REQUIRE(n_points == (pt_hi > 3 ? pt_hi - 3 : 0), "nontrivial evaluation points: incorrect length");
	BN_REQUIRE(n_points <= (uint32_t)bn::kHalMaxPts && total <= 64, "too many evaluation points");
	BN_REQUIRE(n_points == 0 || h_points, "null argument");
	if (total == 0) return BN_OK;

	const uint64_t full = (uint64_t)1 << n_vars, half = full >> 1;
	// ---- Transparent multilinears: partial evaluation at the query into scratch
	uint32_t n_tr = 0;
	for (uint32_t k = 0; k < n_mls; k++) {
		BN_REQUIRE(mls[k].kind == BN_HAL_ML_FOLDED || mls[k].kind == BN_HAL_ML_TRANSPARENT, "unknown multilinear kind");
		BN_REQUIRE(mls[k].d_evals || mls[k].len == 0, "multilinear without evaluations");
		if (mls[k].kind == BN_HAL_ML_TRANSPARENT) n_tr++;
	}
	// ---- the general path's plan (below: "rows + compiled circuits"), made before anything is launched so that the context
	// scratch is laid out once: [partial evaluations of the Transparent multilinears | 1 | rows | circuit temporaries]
	bool general_ok = circuit_multipass_applies(ctx, evs[0].composition, half) && total <= 32;
	uint32_t used_mls = 0; // bit k: multilinear k is read by some composition
	// In the launch-bound regime the rows of all evaluation points of a multilinear lie side by side and ONE set of product passes
	// serves every point that shares a composition (all points but infinity); from 2^18 rows on the passes are traffic-bound, the
	// copies that would put the X = 0 / 1 halves beside the other rows cost more than the launches they save, and every point runs
	// its own passes.
	// (measured, a·b·c + a at X = 1, infinity, z at n = 20 -- rows of 8 MiB --: 0.140 ms with the points side by side, 0.187 ms with
	// a product pass per point; at n = 24 the copies of the X = 1 rows cost more than the launches)
	const bool batch_points = half <= ((uint64_t)1 << 19);
	size_t max_temp_elems = 0; // temporaries of the largest plan, in elements
	for (uint32_t e = 0; e < n_evs && general_ok; e++)
		for (const bn_expr *c : {(const bn_expr *)evs[e].composition, (const bn_expr *)evs[e].composition_at_infinity}) {
			for (const bn_step &st : c->steps)
				if (st.kind == BN_STEP_VAR) used_mls |= 1u << st.a;
			const int t = circuit_multipass_sum_temps(c, evs[e].d_eq_ind != nullptr);
			if (t < 0) general_ok = false;
			const uint32_t n_b = (batch_points && c == evs[e].composition) ? evs[e].eval_point_end - evs[e].eval_point_start : 1;
			// (the final sums of all compositions run in ONE launch at the end, so in the launch-bound regime every composition keeps
			// its own temporaries until then: the sum; at traffic-bound sizes the sums collected so far run before temporaries are
			// reused: the maximum)
			if (t > 0) max_temp_elems = batch_points ? max_temp_elems + (size_t)t * n_b * half : std::max(max_temp_elems, (size_t)t * n_b * half);
		}
	uint32_t n_used = 0;
	for (uint32_t k = 0; k < n_mls; k++) n_used += (used_mls >> k) & 1;
	const size_t tr_elems = n_tr ? (size_t)n_tr * full + 1 : 0;
	const size_t row_elems = general_ok ? (size_t)(pt_hi - pt_lo) * n_used * half : 0;
	const size_t temp_elems = general_ok ? max_temp_elems : 0;
	char *scr = nullptr;
	if (tr_elems + row_elems + temp_elems) {
		scr = (char *)bn::ctx_scratch(ctx, (tr_elems + row_elems + temp_elems) * sizeof(f128));
		if (!scr) {
			if (!n_tr) {
				general_ok = false; // (no room for the rows: the interpreter kernel needs none)
			} else {
				scr = (char *)bn::ctx_scratch(ctx, tr_elems * sizeof(f128));
				general_ok = false;
				if (!scr) return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (scratch)");
			}
		}
	}
	if (n_tr) BN_HIP(bn::launch_fill(ctx->stream, scr + (size_t)n_tr * full * sizeof(f128), 1, bn::f128_one()));
	bn::hal_round_args a{};
	a.n_ml = n_mls;
	a.n_ev = n_evs;
	a.pt_lo = pt_lo;
	a.pt_hi = pt_hi;
	a.order = order;
	a.n_vars = n_vars;
	bool all_full = true;
	uint32_t t = 0;
	for (uint32_t k = 0; k < n_mls; k++) {
		if (mls[k].kind == BN_HAL_ML_TRANSPARENT) {
			void *dst = scr + (size_t)t++ * full * sizeof(f128);
			int rc = materialise(ctx, order, n_vars, mls[k], d_tensor_query, query_vars, scr + (size_t)n_tr * full * sizeof(f128), dst);
			if (rc) return rc;
			a.ml[k].evals = (const uint4 *)dst;
			a.ml[k].len = full;
			a.ml[k].suffix = bn::f128_zero();
		} else {
			a.ml[k].evals = (const uint4 *)mls[k].d_evals;
			a.ml[k].len = mls[k].len < full ? mls[k].len : full;
			a.ml[k].suffix = to_f(&mls[k].suffix_eval);
			if (a.ml[k].len != full) all_full = false;
		}
	}
	for (uint32_t p = 0; p < n_points; p++) a.pts[p] = to_f(&h_points[p]);

	f128 *d_acc = ctx->d_result; // 64 accumulator slots (slots 0..63 of the result area)
	ctx->s_clean = false;
	BN_HIP(hipMemsetAsync(d_acc, 0, 64 * sizeof(f128), ctx->stream));

	// ---- the routed shapes: full Folded multilinears, High-to-Low, evaluation points 1 and infinity, every composition a
	// sum of monomials of at most three multilinears (two next to an equality indicator).  Every DISTINCT monomial
	// (x indicator table) is ONE pass of the ComputeLayer's product-sum kernels, which return the sums at both points;
	// the coefficients are applied to the 16-byte sums on the host.
	// all-ones | all-zeros tables of `half` elements, filled once per size and kept in the context
	auto const_tables = [&]() -> int { return hal_const_tables(ctx, half); };
	struct term_job {
		std::vector<uint32_t> vars;
		const void *eq;
		bool at_inf = false; // some composition's form at infinity holds the monomial
	};
	std::vector<term_job> jobs;
	std::vector<std::vector<monomial>> p1(n_evs), pinf(n_evs);
	char *ones = nullptr, *zeros = nullptr;
	// (Low-to-High order: the pairs are interleaved -- the matrix-core kernel reads them with an element stride of two, so the
	// two-factor jobs take the same path from 2^17 points on)
	const bool l2h = order == BN_ORDER_LOW_TO_HIGH;
	bool fast = (!l2h || bn::mfma_applies(ctx->n_cu, half)) && all_full && pt_lo >= 1 && pt_hi <= 3 && n_vars >= 2;
	auto job_of = [&](const std::vector<uint32_t> &vars, const void *eq) -> int {
		for (size_t j = 0; j < jobs.size(); j++)
			if (jobs[j].vars == vars && jobs[j].eq == eq) return (int)j;
		jobs.push_back(term_job{vars, eq, false});
		return (int)jobs.size() - 1;
	};
	for (uint32_t e = 0; e < n_evs && fast; e++) {
		fast = expand_poly(evs[e].composition, p1[e]) && expand_poly(evs[e].composition_at_infinity, pinf[e]);
		for (const auto *pl : {&p1[e], &pinf[e]})
			for (const auto &t : *pl) {
				if (t.vars.size() + (evs[e].d_eq_ind ? 1 : 0) > (l2h ? 2u : 3u)) fast = false; // (one slot is kept for the all-ones factor)
				if (fast) {
					const int j = job_of(t.vars, evs[e].d_eq_ind);
					if (pl == &pinf[e]) jobs[(size_t)j].at_inf = true;
				}
			}
		if (jobs.size() > 15) fast = false;
	}
	if (fast) {
		// A monomial with fewer than two factors is padded with an all-ones table (sum of v = sum of v * 1), and a
		// two-factor job one of whose factors is the same at both points (the indicator, the ones) gets an all-zeros
		// table as that factor's partner, (f + 0) = f: it then has the shape of the bivariate round evaluation and runs on
		// the matrix cores instead of the generic product-sum kernel.
		// The tables are filled once per size and kept in the context (refilling them on every call wrote as many bytes
		// as the job reads).  A monomial without any factor and without an indicator is a constant: its sum over the
		// 2^(n_vars - 1) >= 2 points of the half cube is zero in characteristic 2 -- no pass at all.
		bool need_tables = false;
		for (const auto &j : jobs)
			if (!(j.vars.empty() && !j.eq) && j.vars.size() + (j.eq ? 1 : 0) <= 2 && (j.eq || j.vars.size() < 2)) need_tables = true;
		if (need_tables) {
			{
				int rc = const_tables();
				if (rc) return rc;
			}
			ones = (char *)ctx->hal_const;
			zeros = ones + ctx->hal_const_half * sizeof(f128);
		}
	}
	if (fast) {
		// slots 32 + 2 j, 33 + 2 j: (S_1, S_inf) of job j
		for (size_t j = 0; j < jobs.size(); j++) {
			if (jobs[j].vars.empty() && !jobs[j].eq) continue; // constant monomial: both sums are zero (the slots are)
			if (!l2h && !jobs[j].eq && jobs[j].vars.size() == 1 && half >= ((uint64_t)1 << 18)) {
				// a lone column: its sums at streaming speed (k_xor_sum) -- as a product with the all-ones table it is a pass over twice
				// the bytes, and over its lower half even where no form at infinity holds it (a b + c at n = 24: 0.31 -> 0.2 ms)
				const char *p = (const char *)a.ml[jobs[j].vars[0]].evals;
				BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, p + half * 16, half, d_acc + 32 + 2 * j));
				if (jobs[j].at_inf) {
					BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, p + half * 16, half, d_acc + 33 + 2 * j));
					BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, p, half, d_acc + 33 + 2 * j));
				}
				continue;
			}
			const void *hi[4] = {nullptr, nullptr, nullptr, nullptr}, *lo[4] = {nullptr, nullptr, nullptr, nullptr};
			uint32_t k = 0, shift[4] = {0, 0, 0, 0};
			for (uint32_t v : jobs[j].vars) {
				const char *p = (const char *)a.ml[v].evals;
				hi[k] = l2h ? p + 16 : p + half * 16;
				lo[k] = p;
				shift[k] = l2h ? 1 : 0;
				k++;
			}
			if (jobs[j].eq) hi[k++] = jobs[j].eq; // (lo = NULL: the same factor at both evaluation points)
			while (k < 2) hi[k++] = ones;
			if (k == 2)
				for (uint32_t f = 0; f < 2; f++)
					if (!lo[f]) lo[f] = zeros;
			if (l2h)
				BN_HIP(bn::launch_roundeval_mfma_pair_strided(ctx->stream, ctx->n_cu, hi[0], lo[0], shift[0], hi[1], lo[1], shift[1], half, d_acc + 32 + 2 * j));
			else
				BN_HIP(roundeval_product_routed_pub(ctx, /*scratch_free=*/n_tr == 0, hi, lo, k, half, d_acc + 32 + 2 * j)); // (a * b * eq: routed from 2^20 points)
		}
		std::vector<f128> sums(2 * jobs.size());
		{
			int rc = publish_vals(ctx, d_acc + 32, 1, (uint32_t)sums.size(), 0, 0, sums.data()); // (through the mailbox: no copy, no synchronisation)
			if (rc) return rc;
		}
		uint32_t off = 0;
		for (uint32_t e = 0; e < n_evs; e++) {
			for (uint32_t p = evs[e].eval_point_start; p < evs[e].eval_point_end; p++) {
				f128 v = bn::f128_zero();
				for (const auto &t : (p == 1 ? p1[e] : pinf[e]))
					v ^= bn::mul_host(t.coeff, sums[2 * job_of(t.vars, evs[e].d_eq_ind) + (p - 1)]);
				h_out[off + (p - evs[e].eval_point_start)] = bn_f128{v.lo, v.hi};
			}
			off += evs[e].eval_point_end - evs[e].eval_point_start;
		}
		BN_HIP(hipMemsetAsync(d_acc, 0, 64 * sizeof(f128), ctx->stream));
		ctx->s_clean = true;
		return BN_OK;
	} else if (general_ok) {
		// ---- everything else at throughput sizes: "rows + compiled circuits".  Every multilinear a composition reads is brought
		// to its value at every evaluation point asked for -- e0 / e1 by a gather (nothing at all for a full multilinear in
		// High-to-Low order), e0 + e1 for infinity, e0 + z (e0 + e1) through the nibble-table kernel for a domain point; pairs
		// as the order says, the constant suffix filled in -- as a plain row of 2^(n_vars - 1) elements, ONCE, whatever the
		// number of evaluators.  The composition (at infinity: its leading form) is then compiled into passes of the throughput
		// kernels over those rows (abi_circuit.cpp): products on the bit-sliced element-wise kernel, the outermost sum of
		// products on the matrix cores.  (The interpreter kernel below walked the cube once per point with a scalar tower
		// product per Mul: 170 x slower per point, profiles/r03/hal.jsonl.)
		char *rows_base = scr + tr_elems * sizeof(f128);
		const size_t temps_off = (tr_elems + row_elems) * sizeof(f128);
		// row storage: multilinear by multilinear, inside a multilinear the evaluation points in the order "all but infinity,
		// ascending; infinity last" -- the points of any evaluator that share its composition are then adjacent
		const uint32_t n_pts = pt_hi - pt_lo;
		auto pos_of = [&](uint32_t p) -> uint32_t {
			if (p == 2) return n_pts - 1;
			return p - pt_lo - ((pt_lo <= 2 && p > 2) ? 1u : 0u);
		};
		std::vector<std::vector<const void *>> row(pt_hi, std::vector<const void *>(n_mls, nullptr));
		bn::hal_rows_args ra{};
		ra.order = order;
		ra.half = half;
		ra.n_out = half;
		auto flush_rows = [&]() -> int {
			if (ra.n_jobs) BN_HIP(bn::launch_hal_rows(ctx->stream, ctx->n_cu, ra));
			ra.n_jobs = 0;
			return BN_OK;
		};
		uint32_t ku = 0;
		for (uint32_t k = 0; k < n_mls; k++) {
			if (!((used_mls >> k) & 1)) continue;
			for (uint32_t p = pt_lo; p < pt_hi; p++) {
				bool wanted = false;
				for (uint32_t e = 0; e < n_evs; e++) wanted = wanted || (p >= evs[e].eval_point_start && p < evs[e].eval_point_end);
				if (!wanted) continue;
				char *dst = rows_base + ((size_t)ku * n_pts + pos_of(p)) * half * sizeof(f128);
				if (!batch_points && !l2h && a.ml[k].len == full && p <= 1) {
					row[p][k] = (const char *)a.ml[k].evals + (p ? half * 16 : 0); // the half itself
					continue;
				}
				// (all the rows of the request in one launch, kernels_hal.hip k_hal_rows)
				auto &jb = ra.jobs[ra.n_jobs++];
				jb.evals = a.ml[k].evals;
				jb.out = (uint4 *)dst;
				jb.len = a.ml[k].len;
				jb.suffix = a.ml[k].suffix;
				jb.z = p >= 3 ? a.pts[p - 3] : f128{0, 0};
				jb.point = p;
				row[p][k] = dst;
				if (ra.n_jobs == (uint32_t)bn::kHalMaxRows) {
					int rc = flush_rows();
					if (rc) return rc;
				}
			}
			ku++;
		}
		{
			int rc = flush_rows();
			if (rc) return rc;
		}
		{
			int rc = const_tables(); // (the all-ones row: the sum of a lone row is its inner product with it)
			if (rc) return rc;
		}
		if (!batch_points) {
			// traffic-bound sizes: every point runs its own passes and product-sum kernels straight into its pair of accumulator
			// slots, nothing waits for anything, ONE publish at the end (collecting the final sums for a group launch was measured
			// here too: the temporaries then force a launch + wait per composition, 1.30 -> 1.41 - 1.49 ms at n = 24)
			uint32_t idx = 0;
			for (uint32_t e = 0; e < n_evs; e++)
				for (uint32_t p = evs[e].eval_point_start; p < evs[e].eval_point_end; p++, idx++) {
					const bn_expr *c = p == 2 ? evs[e].composition_at_infinity : evs[e].composition;
					int rc = circuit_multipass_sum(ctx, c, row[p].data(), half, evs[e].d_eq_ind, d_acc + 2 * idx, temps_off);
					if (rc == kCircuitDeclined) return bn::fail(BN_ERR_CORE_LIB, "internal: a planned circuit was declined");
					if (rc) return rc;
				}
			std::vector<f128> direct(total);
			{
				int rc = publish_vals(ctx, d_acc, 2, total, 1, 2, direct.data()); // value i = slot 2 i + slot 2 i + 1, through the mailbox
				if (rc) return rc;
			}
			for (uint32_t i = 0; i < total; i++) h_out[i] = bn_f128{direct[i].lo, direct[i].hi};
			BN_HIP(hipMemsetAsync(d_acc, 0, 64 * sizeof(f128), ctx->stream));
			ctx->s_clean = true;
			return BN_OK;
		}
		// launch-bound sizes: the element-wise passes of every composition once for all the points that share it (the rows lie side
		// by side), the final sums collected; then ALL sums of the request in one launch
		ip_collector col;
		std::vector<f128> vals(total, bn::f128_zero());
		size_t temp_cursor = 0; // elements of the temporaries' region handed out to compositions whose sums have not run yet
		ctx->s_clean = true;    // (the accumulator slots were zeroed above, nothing has touched them since)
		std::vector<uint32_t> base_idx(n_evs);
		{
			uint32_t idx = 0;
			for (uint32_t e = 0; e < n_evs; e++) {
				base_idx[e] = idx;
				idx += evs[e].eval_point_end - evs[e].eval_point_start;
			}
		}
		for (uint32_t e = 0; e < n_evs; e++) {
			const uint32_t s0 = evs[e].eval_point_start, s1 = evs[e].eval_point_end;
			std::vector<uint32_t> pts; // the points that use evs[e].composition, in row order
			for (uint32_t p = s0; p < s1; p++)
				if (p != 2) pts.push_back(p);
			auto run = [&](const bn_expr *c, const std::vector<uint32_t> &ps) -> int {
				std::vector<uint32_t> out_index;
				for (uint32_t p : ps) out_index.push_back(base_idx[e] + (p - s0));
				const int t = circuit_multipass_sum_temps(c, evs[e].d_eq_ind != nullptr);
				const size_t need = t > 0 ? (size_t)t * ps.size() * half : 0;
				if (temp_cursor + need > temp_elems) {
					// the temporaries of the compositions collected so far are about to be reused: their sums run now
					int rc = circuit_ip_run(ctx, col, half, ctx->hal_const, vals.data());
					if (rc) return rc;
					col = ip_collector{};
					temp_cursor = 0;
				}
				int rc = circuit_multipass_collect(ctx, c, row[ps[0]].data(), half, (uint32_t)ps.size(), evs[e].d_eq_ind, temps_off + temp_cursor * sizeof(f128),
				                                   out_index.data(), col);
				if (rc == kCircuitDeclined) return bn::fail(BN_ERR_CORE_LIB, "internal: a planned circuit was declined");
				temp_cursor += need;
				return rc;
			};
			if (batch_points) {
				if (!pts.empty()) {
					int rc = run(evs[e].composition, pts);
					if (rc) return rc;
				}
			} else {
				for (uint32_t p : pts) {
					int rc = run(evs[e].composition, std::vector<uint32_t>{p});
					if (rc) return rc;
				}
			}
			if (s0 <= 2 && 2 < s1) {
				int rc = run(evs[e].composition_at_infinity, std::vector<uint32_t>{2});
				if (rc) return rc;
			}
		}
		{
			int rc = circuit_ip_run(ctx, col, half, ctx->hal_const, vals.data());
			if (rc) return rc;
		}
		for (uint32_t i = 0; i < total; i++) h_out[i] = bn_f128{vals[i].lo, vals[i].hi};
		return BN_OK;
	} else {
		uint32_t off = 0;
		for (uint32_t e = 0; e < n_evs; e++) {
			int rc = ensure_d_steps(evs[e].composition);
			if (rc) return rc;
			rc = ensure_d_steps(evs[e].composition_at_infinity);
			if (rc) return rc;
			a.ev[e].steps = evs[e].composition->d_steps;
			a.ev[e].n_steps = (uint32_t)evs[e].composition->steps.size();
			a.ev[e].steps_inf = evs[e].composition_at_infinity->d_steps;
			a.ev[e].n_steps_inf = (uint32_t)evs[e].composition_at_infinity->steps.size();
			a.ev[e].pt_start = evs[e].eval_point_start;
			a.ev[e].pt_end = evs[e].eval_point_end;
			a.ev[e].eq = (const uint4 *)evs[e].d_eq_ind;
			a.ev[e].out_off = off;
			off += evs[e].eval_point_end - evs[e].eval_point_start;
		}
		BN_HIP(bn::launch_hal_round_evals(ctx->stream, ctx->n_cu, a, d_acc));
	}
	BN_HIP(hipMemcpyAsync(h_out, d_acc, total * sizeof(f128), hipMemcpyDeviceToHost, ctx->stream));
	BN_HIP(hipStreamSynchronize(ctx->stream));
	BN_HIP(hipMemsetAsync(d_acc, 0, 64 * sizeof(f128), ctx->stream));
	ctx->s_clean = true;
	return BN_OK;
}

int bn_hal_fold_multilinear(bn_ctx *ctx, uint32_t order, uint32_t n_vars, const bn_hal_multilinear *ml, const bn_f128 *challenge,
                            const void *d_tensor_query, uint32_t query_vars, void *d_out, uint64_t out_cap, uint64_t *out_len)
{
	BN_REQUIRE(ctx && ml && challenge && out_len && (d_out || out_cap == 0), "null argument");
	BN_ENTER(ctx);
	BN_FLUSH(ctx);
	BN_REQUIRE(order == BN_ORDER_LOW_TO_HIGH || order == BN_ORDER_HIGH_TO_LOW, "unknown evaluation order");
	BN_REQUIRE(n_vars > 0 && n_vars < 40, "folding requires at least a single variable");
	const uint64_t full = (uint64_t)1 << n_vars, half = full >> 1;
	if (ml->kind == BN_HAL_ML_TRANSPARENT) {
		// switchover (sumcheck_folding.rs:58-112, 164-216): partial evaluation at the query that already holds this
		// round's challenge
		BN_REQUIRE(query_vars > 0 && d_tensor_query, "tensor query missing while a multilinear is still transparent");
		BN_REQUIRE(out_cap >= half, "output buffer too small");
		bn_hal_multilinear t = *ml;
		int rc = materialise(ctx, order, n_vars - 1, t, d_tensor_query, query_vars, nullptr, d_out);
		if (rc) return rc;
		*out_len = half;
		return BN_OK;
	}
	BN_REQUIRE(ml->kind == BN_HAL_ML_FOLDED, "unknown multilinear kind");
	const uint64_t len = ml->len < full ? ml->len : full;
	const f128 z = to_f(challenge), sfx = to_f(&ml->suffix_eval);
	const uint64_t n_out = order == BN_ORDER_LOW_TO_HIGH ? (len + 1) / 2 : (len < half ? len : half);
	BN_REQUIRE(out_cap >= n_out, "output buffer too small");
	if (order == BN_ORDER_LOW_TO_HIGH && n_out) {
		const char *a = (const char *)ml->d_evals, *b = (const char *)d_out;
		BN_REQUIRE(b + n_out * 16 <= a || a + len * 16 <= b, "Low-to-High fold: output must not overlap the input");
	}
	prof_scope ps(ctx, BN_PROF_FOLD);
	if (order == BN_ORDER_HIGH_TO_LOW && len == full && d_out == ml->d_evals) {
		// the ComputeLayer shape: in-place fold of the two halves
		BN_HIP(bn::launch_extrapolate_line(ctx->stream, ctx->n_cu, d_out, (const char *)ml->d_evals + half * 16, half, z));
	} else {
		BN_HIP(bn::launch_hal_fold_lerp(ctx->stream, ctx->n_cu, ml->d_evals, len, sfx, order, half, z, d_out, n_out));
	}
	*out_len = n_out;
	return BN_OK;
}

} // extern "C"
