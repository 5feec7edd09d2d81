// binius_amd/csrc/abi_circuit.cpp -- arbitrary ArithCircuit compositions on the throughput kernels.
//
// The reference compiles any ArithCircuit into a batched packed evaluator (crates/fast_compute/src/arith_circuit.rs:184-478:
// straight-line steps over batches of packed elements, used by sum_composition_evals, layer.rs:797-846, and
// compute_composite, :552-593).  Here a circuit is compiled -- on the host, per call, a few microseconds -- into a
// straight-line program of passes of the kernels that already run at throughput:
//
//     Mul(x, y)        one pass of the bit-sliced element-wise product (kernels_mul9.hip: 224 products per wave-batch,
//                      ~11 wave-instructions per product against ~3000 lane-operations of the scalar tower recursion)
//     Add(x, y)        one streaming XOR pass (kernels_stream.hip)
//     Mul(const, x)    one pass of the nibble-table constant multiplication (ctable.hpp)
//     Pow(x, e)        square-and-multiply over product passes
//     constants        folded on the host; a constant that has to meet a row element-wise is filled into a temporary
//
// with temporaries in context scratch, released after their last use (a circuit of depth d needs O(d) of them, not one per
// step).  A SUM over the rows (sum_composition_evals, the old HAL's round evaluations) never materialises its last level:
// sum(Add(x, y)) = sum(x) + sum(y), and sum(Mul(x, y)) is one pass of the product-SUM kernels -- the GF(2) Gram products on
// the matrix cores from 2^17 points (gram.hpp), the 9-lane kernel below -- so a sum of monomials costs (degree - 2) product
// passes + one Gram pass per monomial.  Every pass is VALU- or HBM-bound at its own roofline; the
// intermediate traffic is what the element-wise product kernel moves anyway.
//
// The scalar interpreter kernels (kernels_roundeval.hip circuit_eval_dev, kernels_hal.hip) remain as the fallback for rows
// too short to fill a wave-batch and for circuits this planner declines (more than kMaxTemps live temporaries).
#include <algorithm>

#include "abi_common.hpp"
#include "hostmul.hpp"

namespace bnabi {

namespace {
constexpr uint32_t kMaxTemps = 24;

struct node {
	enum { NONE, CONST, EXT, TEMP } kind = NONE;
	f128 c{0, 0};            // CONST
	const void *ext = nullptr; // EXT: a caller's row
	int temp = -1;           // TEMP: index of the temporary
};

struct planner {
	bn_ctx *ctx;
	const bn_expr *e;
	const void *const *rows;
	uint64_t n;
	char *scr = nullptr;      // base of the temporaries
	uint32_t n_temps = 0;     // temporaries in use at the same time, at most
	std::vector<int> free_list;
	std::vector<node> val;
	std::vector<int> last_use; // step index of the last reader (as a materialised operand)
	std::vector<int> owner;    // temp -> step that owns it (-1: free)
	bool dry = true;          // first pass: count temporaries only
	void *final_out = nullptr; // compute_composite: the root's destination
	int root = -1;
	const void *ones = nullptr;

	int alloc_temp(int step)
	{
		int t;
		if (!free_list.empty()) {
			t = free_list.back();
			free_list.pop_back();
		} else {
			t = (int)n_temps++;
			owner.push_back(-1);
		}
		owner[t] = step;
		return t;
	}
	void release_dead(int step)
	{
		for (size_t t = 0; t < owner.size(); t++)
			if (owner[t] >= 0 && last_use[owner[t]] <= step && owner[t] != root) {
				owner[t] = -1;
				free_list.push_back((int)t);
			}
	}
	void *temp_ptr(int t) const { return scr + (size_t)t * n * sizeof(f128); }
	const void *ptr_of(const node &v) const { return v.kind == node::EXT ? v.ext : temp_ptr(v.temp); }

	// a CONST that has to meet a row element-wise
	int materialise_const(node &v, int step)
	{
		const int t = alloc_temp(step);
		if (!dry) BN_HIP(bn::launch_fill(ctx->stream, temp_ptr(t), n, v.c));
		v.kind = node::TEMP;
		v.temp = t;
		return BN_OK;
	}
	void *dest(int step, node &r)
	{
		r.kind = node::TEMP;
		if (step == root && final_out) {
			r.temp = -2; // the caller's output buffer
			return final_out;
		}
		r.temp = alloc_temp(step);
		return dry ? nullptr : temp_ptr(r.temp);
	}
	const void *src(const node &v) const { return v.kind == node::TEMP && v.temp == -2 ? final_out : ptr_of(v); }
	// step `i` is the same array as `x` (x + 0, 1 * x, x^1): the temporary lives as long as its last alias is read
	void alias(node &r, const node &x, int i)
	{
		r = x;
		if (x.kind == node::TEMP && x.temp >= 0 && i >= 0) {
			const int o = owner[x.temp];
			if (o >= 0 && last_use[o] < last_use[i]) last_use[o] = last_use[i];
		}
	}

	int mul_nodes(node x, node y, node &r, int step)
	{
		if (x.kind == node::CONST && y.kind == node::CONST) {
			r.kind = node::CONST;
			r.c = bn::mul_host(x.c, y.c);
			return BN_OK;
		}
		if (x.kind == node::CONST) std::swap(x, y);
		if (y.kind == node::CONST) {
			if (y.c == bn::f128_zero()) {
				r.kind = node::CONST;
				r.c = bn::f128_zero();
				return BN_OK;
			}
			if (y.c == bn::f128_one()) {
				alias(r, x, step);
				return BN_OK;
			}
			void *d = dest(step, r);
			if (!dry) BN_HIP(bn::launch_scale_to(ctx->stream, ctx->n_cu, d, src(x), n, y.c));
			return BN_OK;
		}
		void *d = dest(step, r);
		if (!dry) BN_HIP(bn::launch_mul9(ctx->stream, ctx->n_cu, src(x), 1, src(y), 1, 0, d, n));
		return BN_OK;
	}

	int run_steps(bool skip_root_op)
	{
		const auto &st = e->steps;
		val.assign(st.size(), node{});
		free_list.clear();
		owner.clear();
		n_temps = 0;
		for (int i = 0; i < (int)st.size(); i++) {
			if (last_use[i] < 0 && i != root) continue; // dead code
			if (i == root && skip_root_op) break;
			const bn_step &s = st[i];
			node &r = val[i];
			switch (s.kind) {
			case BN_STEP_VAR:
				r.kind = node::EXT;
				r.ext = rows ? rows[s.a] : nullptr; // (rows == nullptr: a dry run that only counts temporaries)
				break;
			case BN_STEP_CONST:
				r.kind = node::CONST;
				r.c = f128{s.cst.lo, s.cst.hi};
				break;
			case BN_STEP_ADD: {
				node x = val[s.a], y = val[s.b];
				if (x.kind == node::CONST && y.kind == node::CONST) {
					r.kind = node::CONST;
					r.c = x.c ^ y.c;
					break;
				}
				if (x.kind == node::CONST) std::swap(x, y);
				if (y.kind == node::CONST) {
					if (y.c == bn::f128_zero()) {
						alias(r, x, i);
						break;
					}
					int rc = materialise_const(y, i);
					if (rc) return rc;
				}
				void *d = dest(i, r);
				if (!dry) BN_HIP(bn::launch_add(ctx->stream, d, src(x), src(y), n));
				break;
			}
			case BN_STEP_MUL: {
				int rc = mul_nodes(val[s.a], val[s.b], r, i);
				if (rc) return rc;
				break;
			}
			case BN_STEP_POW: {
				const node x = val[s.a];
				const uint64_t ex = s.b;
				if (ex == 0) {
					r.kind = node::CONST;
					r.c = bn::f128_one();
					break;
				}
				if (x.kind == node::CONST) {
					f128 acc = bn::f128_one();
					for (int b = 63; b >= 0; b--) {
						acc = bn::mul_host(acc, acc);
						if ((ex >> b) & 1) acc = bn::mul_host(acc, x.c);
					}
					r.kind = node::CONST;
					r.c = acc;
					break;
				}
				if (ex == 1) {
					alias(r, x, i);
					break;
				}
				// left-to-right square-and-multiply; the running value lives in this step's temporaries
				int top = 63;
				while (!((ex >> top) & 1)) top--;
				node acc = x;
				for (int b = top - 1; b >= 0; b--) {
					node sq;
					const bool last = b == 0 && !((ex >> b) & 1);
					int rc = mul_nodes(acc, acc, sq, last ? i : -3 - b);
					if (rc) return rc;
					if (acc.kind == node::TEMP && acc.temp >= 0 && acc.temp != x.temp) { // the previous running value is dead
						owner[acc.temp] = -1;
						free_list.push_back(acc.temp);
					}
					acc = sq;
					if ((ex >> b) & 1) {
						node pr;
						rc = mul_nodes(acc, x, pr, b == 0 ? i : -3 - b);
						if (rc) return rc;
						if (acc.kind == node::TEMP && acc.temp >= 0) {
							owner[acc.temp] = -1;
							free_list.push_back(acc.temp);
						}
						acc = pr;
					}
				}
				r = acc;
				if (r.kind == node::TEMP && r.temp >= 0) owner[r.temp] = i;
				break;
			}
			default: return bn::fail(BN_ERR_INPUT_VALIDATION, "input validation: unknown circuit step kind");
			}
			release_dead(i);
			if (n_temps > kMaxTemps) return kCircuitDeclined; // (the caller falls back to the interpreter)
		}
		return BN_OK;
	}
};

void mark_uses(const bn_expr *e, std::vector<int> &last_use, int root)
{
	last_use.assign(e->steps.size(), -1);
	std::vector<char> live(e->steps.size(), 0);
	if (root >= 0) live[root] = 1;
	for (int i = (int)e->steps.size() - 1; i >= 0; i--) {
		if (!live[i]) continue;
		const bn_step &s = e->steps[i];
		auto use = [&](uint64_t j) {
			live[j] = 1;
			if (last_use[j] < i) last_use[j] = i;
		};
		if (s.kind == BN_STEP_ADD || s.kind == BN_STEP_MUL) {
			use(s.a);
			use(s.b);
		} else if (s.kind == BN_STEP_POW) {
			use(s.a);
		}
	}
	if (root >= 0 && last_use[root] < 0) last_use[root] = (int)e->steps.size(); // the root is read by the caller
}
} // namespace

bool circuit_multipass_applies(const bn_ctx *ctx, const bn_expr *e, uint64_t row_len)
{
	// (BN_CIRCUIT_MULTIPASS=0, read when the context is created, keeps the interpreter kernels; shorter rows: one launch of the
	// interpreter is the faster thing)
	return ctx->circuit_multipass && !e->steps.empty() && row_len >= 1024;
}

// out[i] = expr(rows[0][i], ..., rows[k-1][i]), i < row_len.  scratch_off: bytes of context scratch the caller is using itself.
// kCircuitDeclined = the planner declines; the caller runs the interpreter.
int circuit_multipass_map(bn_ctx *ctx, const bn_expr *e, const void *const *rows, uint64_t row_len, void *out, size_t scratch_off)
{
	planner p{ctx, e, rows, row_len};
	p.root = (int)e->steps.size() - 1;
	p.final_out = out;
	mark_uses(e, p.last_use, p.root);
	p.dry = true;
	int rc = p.run_steps(false);
	if (rc) return rc;
	const uint32_t need = p.n_temps;
	if (need) {
		char *base = (char *)bn::ctx_scratch(ctx, scratch_off + (size_t)need * row_len * sizeof(f128));
		if (!base) return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (circuit temporaries)");
		p.scr = base + scratch_off;
	}
	p.dry = false;
	rc = p.run_steps(false);
	if (rc) return rc;
	const node &r = p.val[p.root];
	if (r.kind == node::CONST) {
		BN_HIP(bn::launch_fill(ctx->stream, out, row_len, r.c));
	} else if (!(r.kind == node::TEMP && r.temp == -2)) {
		// the root is a row itself (or an alias of an earlier temporary): copy
		BN_HIP(hipMemcpyAsync(out, p.src(r), row_len * sizeof(f128), hipMemcpyDeviceToDevice, ctx->stream));
	}
	return BN_OK;
}

namespace {
struct sum_job {
	int x, y; // operand steps (y = -1: the summand itself, times the indicator if there is one)
};
// sum(Add(x, y)) = sum(x) + sum(y): the summands of the root; every summand that is a product of two sub-circuits (and has no
// indicator next to it) is summed by the product-SUM kernels without ever being formed.  Sets up the planner's use map
// (every operand stays alive to the end) and does the dry run.  false: declined.
bool sum_plan(planner &p, bool has_eq, std::vector<sum_job> &jobs, bool &need_ones, uint32_t &n_temps)
{
	const bn_expr *e = p.e;
	std::vector<int> summands;
	{
		std::vector<int> stack{(int)e->steps.size() - 1};
		while (!stack.empty()) {
			const int i = stack.back();
			stack.pop_back();
			const bn_step &s = e->steps[i];
			if (s.kind == BN_STEP_ADD) {
				stack.push_back((int)s.a);
				stack.push_back((int)s.b);
			} else {
				summands.push_back(i);
			}
			// (shared Add nodes expand exponentially -- t_{k+1} = t_k + t_k, forty deep --: the limit holds DURING the expansion)
			if (summands.size() + stack.size() > 16) return false;
		}
	}
	if (summands.size() > 16) return false;
	p.root = -1;
	p.last_use.assign(e->steps.size(), -1);
	const int end = (int)e->steps.size();
	std::vector<char> live(e->steps.size(), 0);
	auto keep = [&](int j) {
		live[j] = 1;
		p.last_use[j] = end;
	};
	jobs.clear();
	for (int i : summands) {
		const bn_step &s = e->steps[i];
		if (s.kind == BN_STEP_MUL && !has_eq) {
			keep((int)s.a);
			keep((int)s.b);
			jobs.push_back(sum_job{(int)s.a, (int)s.b});
		} else {
			keep(i);
			jobs.push_back(sum_job{i, -1});
		}
	}
	for (int i = end - 1; i >= 0; i--) {
		if (!live[i]) continue;
		const bn_step &s = e->steps[i];
		auto use = [&](uint64_t j) {
			live[j] = 1;
			if (p.last_use[j] < i) p.last_use[j] = i;
		};
		if (s.kind == BN_STEP_ADD || s.kind == BN_STEP_MUL) {
			use(s.a);
			use(s.b);
		} else if (s.kind == BN_STEP_POW) {
			use(s.a);
		}
	}
	p.dry = true;
	if (p.run_steps(false) != BN_OK) return false;
	need_ones = false; // (a lone factor is summed by the streaming XOR kernel: no all-ones table)
	n_temps = p.n_temps + (need_ones ? 1 : 0) + 1; // + one spare for a summand with a constant coefficient
	return true;
}
} // namespace

// temporaries of row_len elements circuit_multipass_sum will ask for (-1: declined) -- for callers that lay out the context
// scratch themselves before anything is launched
int circuit_multipass_sum_temps(const bn_expr *e, bool has_eq)
{
	planner p{nullptr, e, nullptr, 0};
	std::vector<sum_job> jobs;
	bool need_ones = false;
	uint32_t n_t = 0;
	if (e->steps.empty() || !sum_plan(p, has_eq, jobs, need_ones, n_t)) return -1;
	return (int)n_t;
}

// d_slots[0] ^ d_slots[1] ^= sum_i expr(rows[.][i]) * (eq ? eq[i] : 1).  The slots are raw accumulators of the product-sum
// kernels (both must be XORed by the reader).
int circuit_multipass_sum(bn_ctx *ctx, const bn_expr *e, const void *const *rows, uint64_t row_len, const void *eq, f128 *d_slots, size_t scratch_off,
                          const void *ones_table)
{
	planner p{ctx, e, rows, row_len};
	std::vector<sum_job> jobs;
	bool need_ones = false;
	uint32_t n_t = 0;
	if (!sum_plan(p, eq != nullptr, jobs, need_ones, n_t)) return kCircuitDeclined;
	// (a caller that already keeps data in the scratch below scratch_off sized it for the temporaries as well -- a growth here would
	// free that data: declined, the interpreter answers)
	if (scratch_off && scratch_off + (size_t)n_t * row_len * sizeof(f128) > ctx->scratch_bytes) return kCircuitDeclined;
	char *base = (char *)bn::ctx_scratch(ctx, scratch_off + (size_t)n_t * row_len * sizeof(f128));
	if (!base) return bn::fail(BN_ERR_ALLOC, "allocation error: allocator is out of memory (circuit temporaries)");
	p.scr = base + scratch_off;
	// (ones_table: the caller keeps an all-ones table of row_len elements around -- no fill per call)
	const char *ones = need_ones ? (ones_table ? (const char *)ones_table : p.scr + (size_t)p.n_temps * row_len * sizeof(f128)) : nullptr;
	char *spare = p.scr + (size_t)(n_t - 1) * row_len * sizeof(f128);
	if (ones && !ones_table) BN_HIP(bn::launch_fill(ctx->stream, const_cast<char *>(ones), row_len, bn::f128_one()));
	p.dry = false;
	int rc = p.run_steps(false);
	if (rc) return rc;
	for (const auto &j : jobs) {
		// coefficient and the (at most two) non-constant factors of the job
		f128 c = bn::f128_one();
		const void *f[2] = {nullptr, nullptr};
		int nf = 0;
		auto take = [&](const node &v) {
			if (v.kind == node::CONST)
				c = bn::mul_host(c, v.c);
			else
				f[nf++] = p.src(v);
		};
		take(p.val[j.x]);
		if (j.y >= 0)
			take(p.val[j.y]);
		else if (eq)
			f[nf++] = eq;
		if (c == bn::f128_zero()) continue;
		if (nf == 0) {
			// a constant summand: c times (number of rows mod 2)
			if (row_len & 1) {
				BN_HIP(bn::launch_fill(ctx->stream, spare, 1, c));
				BN_HIP(bn::launch_add_assign(ctx->stream, d_slots, spare, 1));
			}
			continue;
		}
		if (!(c == bn::f128_one())) { // sum c x y = sum (c x) y: one constant-multiplication pass
			BN_HIP(bn::launch_scale_to(ctx->stream, ctx->n_cu, spare, f[0], row_len, c));
			f[0] = spare;
		}
		if (nf == 1) {
			BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, f[0], row_len, d_slots));
			continue;
		}
		BN_HIP(bn::launch_sum_product(ctx->stream, ctx->n_cu, f, 2, row_len, d_slots));
	}
	return BN_OK;
}

// ---- the same plan for SEVERAL evaluation points at once, its final sums collected instead of launched ------------------------
// rows[v] is a super-row: n_b segments of `seg` elements, segment b = variable v at the b-th evaluation point (abi_hal.cpp lays the
// rows of a request out that way).  The element-wise passes of the plan run ONCE over the super-rows (n_b * seg elements per
// launch); every final sum -- a product of two rows, a lone row, a row times the indicator -- becomes one inner-product request per
// point (segment b of each factor; the indicator is the same at every point), with the summand's constant coefficient left for
// the host.  circuit_ip_run then computes ALL requests of a call -- every evaluator, every point -- in one launch of the group
// kernel (kernels_group.hip, kind 2: two inner products per job).  An `a·b·c + a` at three points is rows + two product passes
// + one launch instead of thirteen launches.
int circuit_multipass_collect(bn_ctx *ctx, const bn_expr *e, const void *const *rows, uint64_t seg, uint32_t n_b, const void *eq, size_t scratch_off,
                              const uint32_t *out_index, ip_collector &col)
{
	const uint64_t n = (uint64_t)n_b * seg;
	planner p{ctx, e, rows, n};
	std::vector<sum_job> jobs;
	bool need_ones = false;
	uint32_t n_t = 0;
	if (!sum_plan(p, eq != nullptr, jobs, need_ones, n_t)) return kCircuitDeclined;
	if (scratch_off + (size_t)n_t * n * sizeof(f128) > ctx->scratch_bytes) return kCircuitDeclined; // (the caller sized the scratch: never grown here)
	p.scr = (char *)ctx->scratch + scratch_off;
	p.dry = false;
	int rc = p.run_steps(false);
	if (rc) return rc;
	for (const auto &j : jobs) {
		f128 c = bn::f128_one();
		const void *f[2] = {nullptr, nullptr};
		bool is_eq[2] = {false, false};
		int nf = 0;
		auto take = [&](const node &v) {
			if (v.kind == node::CONST)
				c = bn::mul_host(c, v.c);
			else
				f[nf++] = p.src(v);
		};
		take(p.val[j.x]);
		if (j.y >= 0) {
			take(p.val[j.y]);
		} else if (eq) {
			is_eq[nf] = true;
			f[nf++] = eq;
		}
		if (c == bn::f128_zero()) continue;
		for (uint32_t b = 0; b < n_b; b++) {
			if (nf == 0) {
				if (seg & 1) col.consts.push_back({out_index[b], c}); // a constant summand: c times (number of rows mod 2)
				continue;
			}
			ip_job q;
			q.a = is_eq[0] ? f[0] : (const char *)f[0] + (size_t)b * seg * sizeof(f128);
			q.b = nf == 2 ? (is_eq[1] ? f[1] : (const char *)f[1] + (size_t)b * seg * sizeof(f128)) : nullptr;
			q.coeff = c;
			q.out = out_index[b];
			col.jobs.push_back(q);
		}
	}
	return BN_OK;
}

// values[q.out] ^= q.coeff * sum_j q.a[j] q.b[j] (q.b == null: the all-ones row `ones`) for every collected request, n elements each;
// as many launches of the group kernel as 32 jobs of two products each take (one, for every request seen so far)
int circuit_ip_run(bn_ctx *ctx, const ip_collector &col, uint64_t n, const void *ones, f128 *values)
{
	(void)ones;
	for (const auto &c : col.consts) values[c.first] ^= c.second;
	// At traffic-bound sizes the sum of a LONE row is a streaming XOR (kernels_stream.hip k_xor_sum: 16 B per element at the copy
	// rate); as a job of the group kernel it would run at the Gram products' rate for nothing (measured: a·b·c + a at n = 24
	// 1.30 -> 1.41 ms with every final sum in the group launch).  Their results wait in accumulator slots 32 .. and come back
	// through the mailbox in one publish.
	std::vector<ip_job> jobs;
	std::vector<ip_job> lone;
	for (const auto &q : col.jobs) {
		if (!q.b && n >= ((uint64_t)1 << 20) && lone.size() < 32)
			lone.push_back(q);
		else
			jobs.push_back(q);
	}
	if (!lone.empty()) {
		if (!ctx->s_clean) {
			BN_HIP(hipMemsetAsync(ctx->d_result, 0, 64 * sizeof(f128), ctx->stream));
			ctx->s_clean = true;
		}
		for (size_t i = 0; i < lone.size(); i++) BN_HIP(bn::launch_xor_sum(ctx->stream, ctx->n_cu, lone[i].a, n, ctx->d_result + 32 + i));
		std::vector<f128> got(lone.size());
		ctx->mirror.valid = false;
		int rc = publish_vals(ctx, ctx->d_result + 32, 1, (uint32_t)lone.size(), 0, 1, got.data());
		if (rc) return rc;
		BN_HIP(hipMemsetAsync(ctx->d_result + 32, 0, lone.size() * sizeof(f128), ctx->stream));
		for (size_t i = 0; i < lone.size(); i++)
			values[lone[i].out] ^= (lone[i].coeff == bn::f128_one()) ? got[i] : bn::mul_host(lone[i].coeff, got[i]);
	}
	size_t at = 0;
	while (at < jobs.size()) {
		// (at most 32 jobs per launch: their 64 sums come back through the 64 value slots of the context's mailbox)
		constexpr uint32_t kIpJobsPerLaunch = 32;
		bn::group_job gj[kIpJobsPerLaunch];
		uint32_t nj = 0, n_ip = 0;
		const size_t first = at;
		while (at < jobs.size() && nj < kIpJobsPerLaunch) {
			bn::group_job &g = gj[nj];
			g = bn::group_job{};
			g.kind = 2;
			g.n = n;
			g.slot = 2 * nj;
			g.x0[0] = jobs[at].a;
			g.x0[1] = jobs[at].b; // (null: the kernel stages the all-ones row itself)
			at++;
			n_ip++;
			if (at < jobs.size()) {
				g.x1[0] = jobs[at].a;
				g.x1[1] = jobs[at].b;
				at++;
				n_ip++;
			}
			nj++;
		}
		if (!ctx->s_clean) {
			BN_HIP(hipMemsetAsync(ctx->d_result, 0, 64 * sizeof(f128), ctx->stream));
			ctx->s_clean = true;
		}
		ctx->mirror.valid = false;
		const uint64_t seq = ++ctx->mail_seq;
		{
			prof_scope ps(ctx, BN_PROF_ROUND_EVAL_MFMA);
			const int rc_a = group_res_alloc(ctx);
			if (rc_a) return rc_a;
			bn::group_tables *h_tb = (bn::group_tables *)ctx->grp.h_tables;
			const bn::group_tables *d_tb = (const bn::group_tables *)ctx->grp.d_tables;
			const hipError_t e = bn::launch_group(ctx->stream, ctx->n_cu, gj, nj, 2 * nj, ctx->d_result, ctx->d_mail, ctx->d_mail, ctx->d_ticket, seq, h_tb->jobs, d_tb->jobs);
			if (e != hipSuccess) return bn::hip_fail(e, "launch_group (inner products of compiled circuits)");
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
		for (uint32_t i = 0; i < n_ip; i++) {
			f128 v;
			v.lo = __atomic_load_n(&ctx->h_mail[i].lo, __ATOMIC_RELAXED);
			v.hi = __atomic_load_n(&ctx->h_mail[i].hi, __ATOMIC_RELAXED);
			const ip_job &q = jobs[first + i];
			values[q.out] ^= (q.coeff == bn::f128_one()) ? v : bn::mul_host(q.coeff, v);
		}
	}
	return BN_OK;
}

} // namespace bnabi
