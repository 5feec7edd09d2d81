// tests/cpp/conformance.cpp -- C++ ports of the reference's backend-conformance tests
// (generic functions in crates/compute_test_utils/src/layer.rs, instantiated per backend in
// crates/compute/tests/layer.rs) against the C++ host mirror binius_amd/host/compute_layer.hpp.
// Expected values come from the oracle (oracle/*.c, linked in here as the checker only).
// Run by tests/test_gpu_cpp_conformance.py on the GPU box; exit code 0 = all passed.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../binius_amd/host/callers.hpp"
#include "../../binius_amd/host/fri.hpp"
#include "../../binius_amd/host/hal_backend.hpp"
#include "../../binius_amd/host/merkle.hpp"
#include "../../binius_amd/host/sumcheck.hpp"
extern "C" {
#include "../../oracle/hal_ref.h"
#include "../../oracle/layer_ref.h"
#include "../../oracle/merkle_ref.h"
#include "../../oracle/ntt_ref.h"
#include "../../oracle/sumcheck_ref.h"
}

using namespace binius_amd;

static int g_fail = 0;
#define CHECK(cond)                                                                     \
	do {                                                                                \
		if (!(cond)) {                                                                  \
			std::printf("    CHECK failed: %s  (%s:%d)\n", #cond, __FILE__, __LINE__); \
			g_fail++;                                                                   \
			return;                                                                     \
		}                                                                               \
	} while (0)

static std::vector<B128> random_vec(uint64_t seed, size_t n)
{
	std::vector<B128> v(n);
	ref_splitmix_fill(seed, reinterpret_cast<uint64_t *>(v.data()), 2 * n);
	return v;
}
[[maybe_unused]] static const ref_b128 *R(const std::vector<B128> &v) { return reinterpret_cast<const ref_b128 *>(v.data()); }
static ref_b128 *R(std::vector<B128> &v) { return reinterpret_cast<ref_b128 *>(v.data()); }
static ref_b128 r1(B128 x) { return ref_b128{x.lo, x.hi}; }
static B128 b1(ref_b128 x) { return B128(x.lo, x.hi); }
static FSlice C(const FSliceMut &m) { return ComputeMemory::as_const(m); }

struct Env {
	ComputeHolder holder{1 << 16, 1 << 20};
};

// layer.rs:792-825
static void test_copy_host_device(Env &e)
{
	ComputeData d = e.holder.to_data();
	auto h1 = random_vec(1, 128);
	std::vector<B128> h2(128);
	FSliceMut d1 = d.dev_alloc.alloc(128), d2 = d.dev_alloc.alloc(128);
	d.hal->copy_h2d(h1, d1);
	d.hal->copy_d2d(C(d1), d2);
	d.hal->copy_d2h(C(d2), h2);
	CHECK(h1 == h2);
	bool threw = false;
	try {
		FSliceMut small = d.dev_alloc.alloc(8);
		d.hal->copy_h2d(h1, small);
	} catch (const Error &err) {
		threw = err.kind() == Error::InputValidation;
	}
	CHECK(threw);
}

// alloc.rs:123-158
static void test_bump_allocator(Env &e)
{
	ComputeData d = e.holder.to_data();
	FSliceMut buf = d.dev_alloc.alloc(256);
	DeviceBumpAllocator bump(buf);
	CHECK(bump.alloc(100).len() == 100);
	CHECK(bump.alloc(100).len() == 100);
	bool oom = false;
	try {
		bump.alloc(100);
	} catch (const Error &err) {
		oom = err.kind() == Error::Alloc;
	}
	CHECK(oom);
	CHECK(bump.capacity() == 56);
	DeviceBumpAllocator sub = bump.subscope_allocator();
	CHECK(sub.alloc(56).len() == 56);
}

// layer.rs:827-848
static void test_log_chunks_range(Env &)
{
	std::vector<KernelMemMap> maps = {KernelMemMap::chunked(FSlice{(void *)0x1000, 256}, 4),
	                                  KernelMemMap::chunked_mut(FSliceMut{(void *)0x9000, 256}, 6), KernelMemMap::local(8)};
	size_t s = 99, en = 99;
	CHECK(KernelMemMap::log_chunks_range(maps, s, en));
	CHECK(s == 0 && en == 2);
}

// compute_test_utils/src/layer.rs:22-72
static void test_tensor_expand(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t n_vars = 8, log_n = 2;
	std::vector<B128> buf(1 << n_vars);
	auto head = random_vec(20, 1 << log_n);
	for (size_t i = 0; i < head.size(); i++) buf[i] = head[i];
	auto coords = random_vec(21, n_vars - log_n);
	FSliceMut dev = d.dev_alloc.alloc(buf.size());
	d.hal->copy_h2d(buf, dev);
	d.hal->execute([&](ComputeLayerExecutor &exec) {
		exec.tensor_expand(log_n, coords, dev);
		return std::vector<B128>{};
	});
	std::vector<B128> got(buf.size());
	d.hal->copy_d2h(C(dev), got);
	ref_tensor_expand(R(buf), buf.size(), log_n, R(coords), coords.size(), 1);
	CHECK(got == buf);
}

// layer.rs:74-125 (a at the B16 level) and every other level
static void test_inner_product(Env &e)
{
	ComputeData d = e.holder.to_data();
	for (size_t level : {0u, 3u, 4u, 5u, 6u, 7u}) {
		const size_t n_b = 1 << 8, n_a = n_b >> (7 - level);
		auto a = random_vec(40 + level, n_a), b = random_vec(50, n_b);
		FSliceMut da = d.dev_alloc.alloc(n_a), db = d.dev_alloc.alloc(n_b);
		d.hal->copy_h2d(a, da);
		d.hal->copy_h2d(b, db);
		auto res = d.hal->execute([&](ComputeLayerExecutor &exec) { return std::vector<B128>{exec.inner_product(SubfieldSlice(C(da), level), C(db))}; });
		ref_b128 want;
		CHECK(ref_inner_product(R(a), n_a, (int)level, R(b), n_b, &want) == 0);
		CHECK(res[0] == b1(want));
	}
}

// layer.rs:127-233: eq_ind_partial_eval + inner products == MLE evaluation, through `join`
static void test_multilinear_evaluations(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t n_vars = 8;
	auto point = random_vec(30, n_vars);
	auto m1 = random_vec(31, (1 << n_vars) >> (7 - 4)), m2 = random_vec(32, (1 << n_vars) >> (7 - 5));
	FSliceMut d1 = d.dev_alloc.alloc(m1.size()), d2 = d.dev_alloc.alloc(m2.size());
	d.hal->copy_h2d(m1, d1);
	d.hal->copy_h2d(m2, d2);
	FSliceMut eq = ops::eq_ind_partial_eval(*d.hal, d.dev_alloc, point);
	std::vector<B128> eq_h(1 << n_vars), exp(1 << n_vars);
	d.hal->copy_d2h(C(eq), eq_h);
	exp[0] = B128::ONE();
	ref_tensor_expand(R(exp), exp.size(), 0, R(point), n_vars, 1);
	CHECK(eq_h == exp);
	auto res = d.hal->execute([&](ComputeLayerExecutor &exec) {
		auto pr = exec.join([&](ComputeLayerExecutor &x) { return x.inner_product(SubfieldSlice(C(d1), 4), C(eq)); },
		                    [&](ComputeLayerExecutor &x) { return x.inner_product(SubfieldSlice(C(d2), 5), C(eq)); });
		return std::vector<B128>{pr.first, pr.second};
	});
	ref_b128 w1, w2;
	ref_inner_product(R(m1), m1.size(), 4, R(exp), exp.size(), &w1);
	ref_inner_product(R(m2), m2.size(), 5, R(exp), exp.size(), &w2);
	CHECK(res[0] == b1(w1) && res[1] == b1(w2));
}

// layer.rs:329-410
static void test_inner_product_using_kernel_accumulator(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t log_len = 8, n = 1 << log_len;
	auto a = random_vec(94, n), b = random_vec(95, n);
	FSliceMut da = d.dev_alloc.alloc(n), db = d.dev_alloc.alloc(n);
	d.hal->copy_h2d(a, da);
	d.hal->copy_h2d(b, db);
	ExprEval eval = d.hal->compile_expr(ArithCircuit::var(0) * ArithCircuit::var(1));
	auto res = d.hal->execute([&](ComputeLayerExecutor &exec) {
		return exec.accumulate_kernels(
		    [&](KernelExecutor &ke, size_t log_chunks, std::vector<KernelBuffer> &bufs) {
			    const size_t log_chunk = log_len - log_chunks;
			    KernelValue acc = ke.decl_value(B128::ZERO());
			    SlicesBatch<KSlice> rows({bufs[0].to_ref(), bufs[1].to_ref()}, (size_t)1 << log_chunk);
			    ke.sum_composition_evals(rows, eval, B128::ONE(), acc);
			    return std::vector<KernelValue>{acc};
		    },
		    {KernelMemMap::chunked(C(da), 3), KernelMemMap::chunked(C(db), 3)});
	});
	ref_b128 want;
	ref_inner_product(R(a), n, 7, R(b), n, &want);
	CHECK(res.size() == 1 && res[0] == b1(want));
}

// layer.rs:412-501
static void test_kernel_add(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t log_len = 10, n = 1 << log_len;
	auto a = random_vec(92, n), b = random_vec(93, n);
	FSliceMut da = d.dev_alloc.alloc(n), db = d.dev_alloc.alloc(n);
	d.hal->copy_h2d(a, da);
	d.hal->copy_h2d(b, db);
	ExprEval eval = d.hal->compile_expr(ArithCircuit::var(0));
	auto res = d.hal->execute([&](ComputeLayerExecutor &exec) {
		return exec.accumulate_kernels(
		    [&](KernelExecutor &ke, size_t log_chunks, std::vector<KernelBuffer> &bufs) {
			    const size_t log_chunk = log_len - log_chunks;
			    ke.add(log_chunk, bufs[0].to_ref(), bufs[1].to_ref(), bufs[2].as_mut());
			    KernelValue acc = ke.decl_value(B128::ZERO());
			    SlicesBatch<KSlice> rows({bufs[2].to_ref()}, (size_t)1 << log_chunk);
			    ke.sum_composition_evals(rows, eval, B128::ONE(), acc);
			    return std::vector<KernelValue>{acc};
		    },
		    {KernelMemMap::chunked(C(da), 3), KernelMemMap::chunked(C(db), 3), KernelMemMap::local(log_len)});
	});
	B128 want;
	for (size_t i = 0; i < n; i++) want += a[i] + b[i];
	CHECK(res[0] == want);
}

// layer.rs:503-570
static void test_fri_fold(Env &e)
{
	for (size_t log_batch : {0u, 4u}) {
		ComputeData d = e.holder.to_data();
		const size_t log_len = 10, n_fold = 2, tw_level = 4; // FSub = B16 as in the reference test
		AdditiveNTT ntt(*d.hal, tw_level, log_len);
		auto data = random_vec(110, (size_t)1 << (log_len + log_batch));
		auto challenges = random_vec(111, log_batch + n_fold);
		const size_t out_len = (size_t)1 << (log_len - n_fold);
		FSliceMut din = d.dev_alloc.alloc(data.size()), dout = d.dev_alloc.alloc(out_len);
		d.hal->copy_h2d(data, din);
		d.hal->execute([&](ComputeLayerExecutor &exec) {
			exec.fri_fold(ntt, log_len, log_batch, challenges, C(din), dout);
			return std::vector<B128>{};
		});
		std::vector<B128> got(out_len), want(out_len);
		d.hal->copy_d2h(C(dout), got);
		std::vector<uint64_t> s(REF_NTT_MAX_DIM * REF_NTT_MAX_DIM);
		CHECK(ref_ntt_s_evals((int)tw_level, (int)log_len, s.data()) == 0);
		for (size_t i = 0; i < s.size(); i++) CHECK(s[i] == ntt.s_evals()[i]);
		CHECK(ref_fold_interleaved(s.data(), (int)tw_level, (int)log_len, (int)log_len, (int)log_batch, R(challenges), challenges.size(),
		                           R(data), data.size(), R(want), out_len) == 0);
		CHECK(got == want);
	}
}

// layer.rs:572-724
static void test_left_right_fold(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t level = 4, log_evals = 4, log_q = 1; // evals 2^4 B16 elements inside B128s, query 2^1
	auto mat = random_vec(60, std::max<size_t>(1, ((size_t)1 << log_evals) >> (7 - level)));
	auto vec = random_vec(61, 1 << log_q);
	const size_t out_len = (size_t)1 << (log_evals - log_q);
	FSliceMut dm = d.dev_alloc.alloc(mat.size()), dv = d.dev_alloc.alloc(vec.size()), dl = d.dev_alloc.alloc(out_len), dr = d.dev_alloc.alloc(out_len);
	d.hal->copy_h2d(mat, dm);
	d.hal->copy_h2d(vec, dv);
	d.hal->execute([&](ComputeLayerExecutor &exec) {
		exec.fold_left(SubfieldSlice(C(dm), level), C(dv), dl);
		exec.fold_right(SubfieldSlice(C(dm), level), C(dv), dr);
		return std::vector<B128>{};
	});
	std::vector<B128> gl(out_len), gr(out_len), wl(out_len), wr(out_len);
	d.hal->copy_d2h(C(dl), gl);
	d.hal->copy_d2h(C(dr), gr);
	CHECK(ref_fold_left(R(mat), mat.size(), (int)level, R(vec), vec.size(), R(wl), out_len) == 0);
	CHECK(ref_fold_right(R(mat), mat.size(), (int)level, R(vec), vec.size(), R(wr), out_len) == 0);
	CHECK(gl == wl && gr == wr);
}

// layer.rs:726-771
static void test_extrapolate_line(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t n = 1 << 10;
	auto e0 = random_vec(10, n), e1 = random_vec(11, n);
	B128 z = random_vec(12, 1)[0];
	FSliceMut d0 = d.dev_alloc.alloc(n), d1 = d.dev_alloc.alloc(n);
	d.hal->copy_h2d(e0, d0);
	d.hal->copy_h2d(e1, d1);
	d.hal->execute([&](ComputeLayerExecutor &exec) {
		exec.extrapolate_line(d0, C(d1), z);
		return std::vector<B128>{};
	});
	std::vector<B128> got(n);
	d.hal->copy_d2h(C(d0), got);
	ref_extrapolate_line(R(e0), R(e1), n, n, r1(z));
	CHECK(got == e0);
	bool threw = false;
	try {
		FSliceMut half = ComputeMemory::slice_mut(d0, 0, n / 2);
		d.hal->execute([&](ComputeLayerExecutor &exec) {
			exec.extrapolate_line(half, C(d1), z);
			return std::vector<B128>{};
		});
	} catch (const Error &err) {
		threw = err.kind() == Error::InputValidation;
	}
	CHECK(threw);
}

// layer.rs:773-826
static void test_compute_composite(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t n = 1 << 10;
	auto a = random_vec(70, n), b = random_vec(71, n);
	FSliceMut da = d.dev_alloc.alloc(n), db = d.dev_alloc.alloc(n), dout = d.dev_alloc.alloc(n);
	d.hal->copy_h2d(a, da);
	d.hal->copy_h2d(b, db);
	ExprEval eval = d.hal->compile_expr(ArithCircuit::var(0) * ArithCircuit::var(1));
	d.hal->execute([&](ComputeLayerExecutor &exec) {
		exec.compute_composite(SlicesBatch<FSlice>({C(da), C(db)}, n), dout, eval);
		return std::vector<B128>{};
	});
	std::vector<B128> got(n), want(n);
	d.hal->copy_d2h(C(dout), got);
	ref_b128_mul_vec(R(a), R(b), R(want), n);
	CHECK(got == want);
}

// layer.rs:828-907
static void test_map_kernels(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t log_len = 10, n = 1 << log_len;
	auto a = random_vec(90, n), b = random_vec(91, n);
	for (auto &x : b) x = B128(x.lo & 1, 0); // b in {0, 1} as in the reference test
	FSliceMut da = d.dev_alloc.alloc(n), db = d.dev_alloc.alloc(n);
	d.hal->copy_h2d(a, da);
	d.hal->copy_h2d(b, db);
	d.hal->execute([&](ComputeLayerExecutor &exec) {
		exec.map_kernels(
		    [&](KernelExecutor &ke, size_t log_chunks, std::vector<KernelBuffer> &bufs) {
			    ke.add_assign(log_len - log_chunks, bufs[1].to_ref(), bufs[0].as_mut());
		    },
		    {KernelMemMap::chunked_mut(da, 0), KernelMemMap::chunked(C(db), 0)});
		return std::vector<B128>{};
	});
	std::vector<B128> got(n);
	d.hal->copy_d2h(C(da), got);
	for (size_t i = 0; i < n; i++) CHECK(got[i] == a[i] + b[i]);
}

// layer.rs:909-960
static void test_pairwise_product_reduce(Env &e)
{
	for (size_t log_n : {1u, 8u}) {
		ComputeData d = e.holder.to_data();
		const size_t n = (size_t)1 << log_n;
		auto x = random_vec(80, n);
		FSliceMut dx = d.dev_alloc.alloc(n);
		d.hal->copy_h2d(x, dx);
		std::vector<FSliceMut> outs;
		for (size_t r = 0; r < log_n; r++) outs.push_back(d.dev_alloc.alloc(n >> (r + 1)));
		d.hal->execute([&](ComputeLayerExecutor &exec) {
			exec.pairwise_product_reduce(C(dx), outs);
			return std::vector<B128>{};
		});
		std::vector<B128> cur = x;
		for (size_t r = 0; r < log_n; r++) {
			std::vector<B128> next(cur.size() / 2), got(cur.size() / 2);
			for (size_t i = 0; i < next.size(); i++) next[i] = b1(ref_b128_mul(r1(cur[2 * i]), r1(cur[2 * i + 1])));
			d.hal->copy_d2h(C(outs[r]), got);
			CHECK(got == next);
			cur = next;
		}
	}
}

// compute_test_utils/src/bivariate_sumcheck.rs:44-262 (transcript replaced by a challenge stream)
static void test_bivariate_sumcheck_prove_verify(Env &e)
{
	const size_t n_vars = 8, m = 8, n_comps = 8;
	ComputeData d = e.holder.to_data();
	std::vector<std::vector<B128>> mls;
	std::vector<FSlice> dev;
	for (size_t j = 0; j < m; j++) {
		mls.push_back(random_vec(0xB1A50000 + j, (size_t)1 << n_vars));
		FSliceMut s = d.dev_alloc.alloc(mls[j].size());
		d.hal->copy_h2d(mls[j], s);
		dev.push_back(C(s));
	}
	std::vector<IndexCompositionBivariate> comps;
	std::vector<uint32_t> flat;
	std::vector<B128> sums;
	for (size_t c = 0; c < n_comps; c++) {
		size_t i = (c * 5 + 1) % m, j = (c * 3 + 2) % m;
		comps.push_back(IndexCompositionBivariate{m, {i, j}});
		flat.push_back((uint32_t)i);
		flat.push_back((uint32_t)j);
		ref_b128 s;
		ref_inner_product(R(mls[i]), mls[i].size(), 7, R(mls[j]), mls[j].size(), &s);
		sums.push_back(b1(s));
	}
	auto stream = random_vec(0xC4A1, n_vars + 1);
	B128 batch_coeff = stream[0];
	std::vector<B128> challenges(stream.begin() + 1, stream.end());
	BivariateSumcheckProver prover(*d.hal, d.dev_alloc, d.host_alloc, n_vars, comps, sums, dev);
	std::vector<B128> got_coeffs;
	B128 running = evaluate_univariate(sums, batch_coeff);
	for (size_t r = 0; r < n_vars; r++) {
		auto rc = prover.execute(batch_coeff);
		CHECK(rc.size() == 3);
		CHECK(rc[0] + (rc[0] + rc[1] + rc[2]) == running); // verifier: P(0) + P(1) == sum
		running = evaluate_univariate(rc, challenges[r]);
		got_coeffs.insert(got_coeffs.end(), rc.begin(), rc.end());
		prover.fold(challenges[r]);
	}
	auto finals = prover.finish();
	// oracle prover on copies
	std::vector<std::vector<B128>> copies = mls;
	std::vector<ref_b128 *> ptrs;
	for (auto &c : copies) ptrs.push_back(R(c));
	std::vector<B128> want_coeffs(3 * n_vars), want_final(m);
	CHECK(ref_bivariate_sumcheck_prove(ptrs.data(), m, (unsigned)n_vars, flat.data(), n_comps, R(sums), r1(batch_coeff), R(challenges),
	                                   R(want_coeffs), R(want_final), 1) == 0);
	CHECK(got_coeffs == want_coeffs);
	CHECK(finals == want_final);
	// final evals == MLE at the reversed challenges (bivariate_sumcheck.rs:257-261)
	std::vector<B128> point(challenges.rbegin(), challenges.rend());
	for (size_t j = 0; j < m; j++) CHECK(finals[j] == b1(ref_mle_evaluate(R(mls[j]), (unsigned)n_vars, R(point))));
	// state machine errors
	bool threw = false;
	try {
		prover.fold(challenges[0]);
	} catch (const SumcheckError &) {
		threw = true;
	}
	CHECK(threw);
}

// crates/ntt/src/tests/ntt_tests.rs: forward == scalar reference, inverse round trip
static void test_ntt(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t log_n = 12;
	AdditiveNTT ntt(*d.hal, 5, log_n);
	std::vector<uint32_t> data((size_t)1 << log_n), want, got(data.size());
	std::vector<uint64_t> w(data.size());
	ref_splitmix_fill(0x0177, w.data(), w.size());
	for (size_t i = 0; i < data.size(); i++) data[i] = (uint32_t)w[i];
	want = data;
	FSliceMut dev = d.dev_alloc.alloc(data.size() / 4);
	bn_ctx *ctx = d.hal->raw_ctx();
	check(bn_copy_h2d(ctx, reinterpret_cast<const bn_f128 *>(data.data()), data.size() / 4, dev.ptr, dev.len()));
	ntt.forward_transform(dev.ptr, 5, NTTShape{0, log_n, 0}, 0, 0, 0);
	check(bn_copy_d2h(ctx, dev.ptr, dev.len(), reinterpret_cast<bn_f128 *>(got.data()), got.size() / 4));
	std::vector<uint64_t> s(REF_NTT_MAX_DIM * REF_NTT_MAX_DIM);
	ref_ntt_s_evals(5, (int)log_n, s.data());
	CHECK(ref_ntt_forward(want.data(), 5, 5, s.data(), (int)log_n, 0, (int)log_n, 0, 0, 0, 0) == 0);
	CHECK(got == want);
	ntt.inverse_transform(dev.ptr, 5, NTTShape{0, log_n, 0}, 0, 0, 0);
	check(bn_copy_d2h(ctx, dev.ptr, dev.len(), reinterpret_cast<bn_f128 *>(got.data()), got.size() / 4));
	CHECK(got == data);
}

// crates/core/src/merkle_tree/tests.rs:13-45, 47-88, 90-102: commit, open every index at every layer,
// verify with the oracle's restatement of verify_opening / verify_layer / verify_vector
static void test_binary_merkle_vcs(Env &e)
{
	ComputeData d = e.holder.to_data();
	const size_t batch = 4, log_len = 5, n = batch << log_len;
	auto data = random_vec(0x3E7, n);
	FSliceMut dd = d.dev_alloc.alloc(n);
	d.hal->copy_h2d(data, dd);
	BinaryMerkleTreeProver prover(*d.hal);
	auto [commitment, tree] = prover.commit(C(dd), batch, d.dev_alloc);
	CHECK(commitment.depth == log_len);
	CHECK(commitment.root == tree.root());
	// verify_vector: the whole tree recomputed by the oracle
	std::vector<Digest> want(2 * ((size_t)1 << log_len) - 1);
	CHECK(ref_merkle_build(reinterpret_cast<const uint8_t *>(data.data()), n, batch, want[0].data()) == 0);
	CHECK(want == tree.inner_nodes());
	for (size_t layer_depth = 0; layer_depth <= log_len; layer_depth++) {
		auto layer = prover.layer(tree, layer_depth);
		CHECK(layer.size() == (size_t)1 << layer_depth);
		for (size_t index = 0; index < (size_t)1 << log_len; index++) {
			auto branch = prover.prove_opening(tree, layer_depth, index);
			CHECK(branch.size() == log_len - layer_depth);
			Digest leaf, top;
			ref_groestl256(reinterpret_cast<const uint8_t *>(data.data() + index * batch), 16 * batch, leaf.data());
			ref_merkle_root_from_branch(leaf.data(), index, branch.empty() ? nullptr : branch[0].data(), (uint32_t)branch.size(), top.data());
			CHECK(top == layer[index >> (log_len - layer_depth)]);
		}
	}
	bool threw = false;
	try {
		prover.commit(ComputeMemory::slice(C(dd), 0, 24), 5, d.dev_alloc);
	} catch (const Error &ex) {
		threw = ex.kind() == Error::InputValidation && std::string(ex.what()).find("IncorrectBatchSize") != std::string::npos;
	}
	CHECK(threw);
	threw = false;
	try {
		prover.commit(ComputeMemory::slice(C(dd), 0, 24), 8, d.dev_alloc);
	} catch (const Error &ex) {
		threw = std::string(ex.what()).find("PowerOfTwoLengthRequired") != std::string::npos;
	}
	CHECK(threw);
}

// crates/core/src/protocols/fri/tests.rs (test_commit_prove_verify_*): commit the interleaved message,
// run every fold round, finalize, open queries -- against the composition of the oracle's NTT, fri_fold
// and Merkle restatements, bit for bit
static void test_fri_commit_fold_query(Env &e)
{
	struct Shape {
		size_t log_dim, log_inv_rate, log_batch;
		std::vector<size_t> arities;
	};
	for (const Shape &sh : {Shape{8, 2, 3, {3, 2, 1}}, Shape{8, 2, 0, {4, 3}}, Shape{6, 1, 2, {}}}) {
		ComputeData d = e.holder.to_data();
		FRIParams p(sh.log_dim, sh.log_inv_rate, sh.log_batch, sh.arities, 3);
		AdditiveNTT ntt(*d.hal, 5, p.rs_log_len());
		BinaryMerkleTreeProver merkle(*d.hal);
		auto message = random_vec(0xF21 + sh.log_dim + sh.log_batch, (size_t)1 << (sh.log_dim + sh.log_batch));
		FSliceMut dmsg = d.dev_alloc.alloc(message.size());
		d.hal->copy_h2d(message, dmsg);
		CommitOutput out = commit_interleaved(*d.hal, d.dev_alloc, p, ntt, merkle, C(dmsg));
		// oracle: repeat, NTT, tree
		std::vector<B128> code;
		for (size_t j = 0; j < ((size_t)1 << sh.log_inv_rate); j++) code.insert(code.end(), message.begin(), message.end());
		CHECK(ref_ntt_forward(code.data(), 5, 5, ntt.s_evals(), (int)p.rs_log_len(), (int)sh.log_batch + 2, (int)p.rs_log_len(), 0, 0, 0,
		                      (int)sh.log_inv_rate) == 0);
		std::vector<B128> got_code(code.size());
		d.hal->copy_d2h(C(out.codeword), got_code);
		CHECK(got_code == code);
		const size_t coset0 = sh.arities.empty() ? sh.log_dim + sh.log_batch : sh.arities[0];
		std::vector<Digest> want_nodes(2 * (code.size() >> coset0) - 1);
		CHECK(ref_merkle_build(reinterpret_cast<const uint8_t *>(code.data()), code.size(), (uint64_t)1 << coset0, want_nodes[0].data()) == 0);
		CHECK(out.committed.inner_nodes() == want_nodes);
		CHECK(out.commitment == want_nodes.back());
		// fold rounds
		FRIFolder folder(*d.hal, p, ntt, merkle, C(out.codeword), out.committed);
		auto challenges = random_vec(0xC4A + sh.log_dim, folder.n_rounds());
		std::vector<std::vector<B128>> codes{code};
		std::vector<B128> pending;
		size_t cur_log_len = p.rs_log_len(), cur_log_batch = sh.log_batch, next_commit = sh.arities.empty() ? 0 : sh.arities[0], n_committed = 0;
		for (size_t r = 1; r <= challenges.size(); r++) {
			auto [has_root, root] = folder.execute_fold_round(d.dev_alloc, challenges[r - 1]);
			pending.push_back(challenges[r - 1]);
			if (r != next_commit) {
				CHECK(!has_root);
				continue;
			}
			const size_t new_log_len = cur_log_len - (pending.size() - cur_log_batch);
			std::vector<B128> nxt((size_t)1 << new_log_len);
			CHECK(ref_fri_fold(ntt.s_evals(), 5, (int)p.rs_log_len(), (int)cur_log_len, (int)cur_log_batch, R(pending), pending.size(),
			                   R(codes.back()), codes.back().size(), R(nxt), nxt.size()) == 0);
			n_committed++;
			const size_t coset = (size_t)1 << (n_committed < sh.arities.size() ? sh.arities[n_committed] : p.n_final_challenges());
			std::vector<Digest> nodes(2 * (nxt.size() / coset) - 1);
			CHECK(ref_merkle_build(reinterpret_cast<const uint8_t *>(nxt.data()), nxt.size(), coset, nodes[0].data()) == 0);
			CHECK(has_root && root == nodes.back());
			CHECK(folder.round_committed().back().second.inner_nodes() == nodes);
			codes.push_back(nxt);
			cur_log_len = new_log_len;
			cur_log_batch = 0;
			pending.clear();
			next_commit = n_committed < sh.arities.size() ? next_commit + sh.arities[n_committed] : 0;
		}
		CHECK(folder.round_committed().size() == sh.arities.size());
		auto [terminate, qp] = folder.finalize();
		CHECK(terminate == codes.back());
		// queries: opened cosets == the oracle's codewords, branches lead to the advertised layers
		auto layers = qp.vcs_optimal_layers();
		auto depths = p.optimal_layer_depths();
		for (size_t index : {(size_t)0, ((size_t)1 << p.index_bits()) - 1, (size_t)0x5A5 & (((size_t)1 << p.index_bits()) - 1)}) {
			auto openings = qp.prove_query(index);
			CHECK(openings.size() == sh.arities.size());
			size_t idx = index;
			for (size_t i = 0; i < openings.size(); i++) {
				const size_t arity = sh.arities[i];
				if (i > 0) idx >>= arity;
				const auto &cw = codes[i];
				CHECK(openings[i].values == std::vector<B128>(cw.begin() + (idx << arity), cw.begin() + ((idx + 1) << arity)));
				size_t log_cw = 0;
				while (((size_t)1 << log_cw) < cw.size()) log_cw++;
				const size_t log_n_cosets = log_cw - arity;
				CHECK(openings[i].branch.size() == log_n_cosets - depths[i]);
				Digest leaf, top;
				ref_groestl256(reinterpret_cast<const uint8_t *>(openings[i].values.data()), 16 * openings[i].values.size(), leaf.data());
				ref_merkle_root_from_branch(leaf.data(), idx, openings[i].branch.empty() ? nullptr : openings[i].branch[0].data(),
				                            (uint32_t)openings[i].branch.size(), top.data());
				CHECK(top == layers[i][idx >> (log_n_cosets - depths[i])]);
			}
		}
	}
	bool threw = false;
	try {
		FRIParams bad(4, 1, 0, {2, 2}, 1);
	} catch (const Error &ex) {
		threw = std::string(ex.what()).find("InvalidFoldAritySequence") != std::string::npos;
	}
	CHECK(threw);
}

// crates/core/src/protocols/prodcheck/prove.rs:24-77 and crates/core/src/ring_switch/eq_ind.rs:123-141
// (compute_test_utils/src/ring_switch.rs) against the oracle's element-wise product, tensor_expand, fold_right
static void test_prodcheck_and_ring_switch_callers(Env &e)
{
	{
		ComputeData d = e.holder.to_data();
		const size_t log_n = 9;
		auto evals = random_vec(0x9C0D, (size_t)1 << log_n);
		FSliceMut dv = d.dev_alloc.alloc(evals.size());
		d.hal->copy_h2d(evals, dv);
		ProductCircuitLayers pcl = ProductCircuitLayers::compute(C(dv), *d.hal, d.dev_alloc);
		CHECK(pcl.layers().size() == log_n);
		std::vector<B128> cur = evals;
		std::vector<std::vector<B128>> want;
		for (size_t i = 0; i < log_n; i++) {
			want.push_back(cur);
			const size_t half = cur.size() / 2;
			std::vector<B128> nxt(half);
			ref_b128_mul_vec(R(cur), R(cur) + half, R(nxt), half);
			cur = nxt;
		}
		for (size_t i = 0; i < log_n; i++) {
			const auto &w = want[log_n - 1 - i];
			CHECK(pcl.layers()[i].len() == w.size());
			std::vector<B128> got(w.size());
			d.hal->copy_d2h(pcl.layers()[i], got);
			CHECK(got == w);
		}
		CHECK(pcl.product() == cur[0]);
	}
	for (size_t kappa : {(size_t)2, (size_t)4, (size_t)7}) {
		ComputeData d = e.holder.to_data();
		const size_t n_vars = 9;
		auto z_vals = random_vec(0x515 + kappa, n_vars);
		auto coeffs = random_vec(0x516 + kappa, (size_t)1 << kappa);
		const B128 mixing = random_vec(0x517, 1)[0];
		RingSwitchEqInd rs(z_vals, coeffs, mixing, kappa);
		auto pre = rs.precompute_values(*d.hal, d.dev_alloc);
		FSlice mle{};
		d.hal->execute([&](ComputeLayerExecutor &exec) {
			mle = rs.multilinear_extension(pre, exec);
			return std::vector<B128>{};
		});
		std::vector<B128> got((size_t)1 << n_vars), ev((size_t)1 << n_vars), want((size_t)1 << n_vars);
		d.hal->copy_d2h(mle, got);
		ev[0] = mixing;
		CHECK(ref_tensor_expand(R(ev), ev.size(), 0, R(z_vals), n_vars, 1) == 0);
		CHECK(ref_fold_right(R(ev), ev.size(), (int)(7 - kappa), R(coeffs), coeffs.size(), R(want), want.size()) == 0);
		CHECK(got == want);
	}
}

// The old HAL driven the way ProverState drives it (prover_state.rs:138-265): a degree-2 composition a * b + c over one
// large-field multilinear, one B32-packed multilinear that switches over after round 1 and one truncated multilinear
// with a constant suffix; every round's evaluations (X = 1, infinity and one interpolation-domain point) and the
// folded state are compared with the oracle's restatement of CpuBackend driven through the same rounds.
static void test_old_hal_prover_state(Env &e)
{
	for (EvaluationOrder order : {EvaluationOrder::LowToHigh, EvaluationOrder::HighToLow}) {
		ComputeData d = e.holder.to_data();
		const size_t n_vars = 9, n = (size_t)1 << n_vars, level = 5, sw = 2;
		auto a = random_vec(0x01DA, n), packed = random_vec(0x01DB, n >> (7 - level)), c = random_vec(0x01DC, 300);
		const B128 c_suffix = random_vec(0x01DD, 1)[0];
		auto zs = random_vec(0x01DE, n_vars + 1);
		FSliceMut da = d.dev_alloc.alloc(a.size()), dp = d.dev_alloc.alloc(packed.size()), dc = d.dev_alloc.alloc(c.size());
		d.hal->copy_h2d(a, da);
		d.hal->copy_h2d(packed, dp);
		d.hal->copy_h2d(c, dc);
		const ArithCircuit comp = ArithCircuit::var(0) * ArithCircuit::var(1) + ArithCircuit::var(2);
		const ArithCircuit lead = ArithCircuit::var(0) * ArithCircuit::var(1);
		Mi355xBackend backend(*d.hal);
		SumcheckEvaluator ev;
		ev.composition = d.hal->compile_expr(comp);
		ev.composition_at_infinity = d.hal->compile_expr(lead);
		ev.eval_point_start = 1;
		ev.eval_point_end = 4;
		ProverState st(backend, d.dev_alloc, order, n_vars,
		               {SumcheckMultilinear::folded(C(da)), SumcheckMultilinear::transparent(SubfieldSlice(C(dp), level), n_vars, sw),
		                SumcheckMultilinear::folded(C(dc), c_suffix)},
		               {zs[n_vars]});
		// the oracle's state
		std::vector<std::vector<B128>> h = {a, packed, c};
		std::vector<ref_hal_multilinear> rm(3);
		rm[0] = ref_hal_multilinear{REF_HAL_ML_FOLDED, 0, R(h[0]), h[0].size(), ref_b128{0, 0}, 0};
		rm[1] = ref_hal_multilinear{REF_HAL_ML_TRANSPARENT, (uint32_t)level, R(h[1]), h[1].size(), ref_b128{0, 0}, (uint32_t)n_vars};
		rm[2] = ref_hal_multilinear{REF_HAL_ML_FOLDED, 0, R(h[2]), h[2].size(), r1(c_suffix), 0};
		size_t rounds_to_switch = sw;
		std::vector<B128> challenges, query;
		const ref_step *cs = reinterpret_cast<const ref_step *>(comp.steps().data()), *ls = reinterpret_cast<const ref_step *>(lead.steps().data());
		for (size_t r = 0; r < n_vars; r++) {
			const size_t nv = n_vars - r;
			auto got = st.calculate_round_evals({ev});
			ref_hal_evaluator re{cs, comp.steps().size(), ls, lead.steps().size(), 1, 4, nullptr};
			std::vector<B128> want(3);
			const ref_b128 pt = r1(zs[n_vars]);
			CHECK(ref_hal_round_evals((int)order, (uint32_t)nv, query.empty() ? nullptr : R(query), (uint32_t)challenges.size(), rm.data(), 3, &re, 1, &pt, 1,
			                          R(want)) == 0);
			CHECK(got.size() == 1 && got[0].evals == want);
			// fold
			const B128 z = zs[r];
			st.fold(z);
			if (order == EvaluationOrder::LowToHigh) challenges.push_back(z); else challenges.insert(challenges.begin(), z);
			const bool transparent = rm[1].kind == REF_HAL_ML_TRANSPARENT;
			if (transparent) {
				query.assign((size_t)1 << challenges.size(), B128());
				query[0] = B128::ONE();
				CHECK(ref_tensor_expand(R(query), query.size(), 0, R(challenges), challenges.size(), 1) == 0);
			}
			for (size_t j = 0; j < 3; j++) {
				if (j == 1 && transparent && rounds_to_switch > 0) {
					rounds_to_switch--;
					continue;
				}
				std::vector<B128> out((size_t)1 << (nv - 1));
				uint64_t n_out = 0;
				CHECK(ref_hal_fold_multilinear((int)order, (uint32_t)nv, &rm[j], r1(z), query.empty() ? nullptr : R(query), (uint32_t)challenges.size(), R(out),
				                               out.size(), &n_out) == 0);
				out.resize(n_out);
				h[j] = out;
				rm[j] = ref_hal_multilinear{REF_HAL_ML_FOLDED, 0, R(h[j]), h[j].size(), rm[j].kind == REF_HAL_ML_FOLDED ? rm[j].suffix_eval : ref_b128{0, 0}, 0};
			}
			if (rm[1].kind == REF_HAL_ML_FOLDED) { // the query is dropped once nothing is transparent (prover_state.rs:182-184)
				query.clear();
				challenges.clear();
			}
			// device state == oracle state
			for (size_t j = 0; j < 3; j++) {
				const auto &m = st.multilinears()[j];
				if (rm[j].kind == REF_HAL_ML_TRANSPARENT) {
					CHECK(m.kind == SumcheckMultilinear::Transparent);
					continue;
				}
				CHECK(m.kind == SumcheckMultilinear::Folded && m.large_field_folded_evals.len_ == h[j].size());
				if (!h[j].empty()) {
					std::vector<B128> dev(h[j].size());
					d.hal->copy_d2h(m.large_field_folded_evals, dev);
					CHECK(dev == h[j]);
				}
			}
		}
		auto fin = st.finish(*d.hal);
		CHECK(fin.size() == 3);
		for (size_t j = 0; j < 3; j++) CHECK(fin[j] == (h[j].empty() ? b1(rm[j].suffix_eval) : h[j][0]));
	}
}

int main()
{
	struct T {
		const char *name;
		std::function<void(Env &)> fn;
	};
	std::vector<T> tests = {
	    {"test_copy_host_device", test_copy_host_device},
	    {"test_bump_allocator", test_bump_allocator},
	    {"test_log_chunks_range", test_log_chunks_range},
	    {"test_generic_single_tensor_expand", test_tensor_expand},
	    {"test_generic_single_inner_product", test_inner_product},
	    {"test_generic_multiple_multilinear_evaluations", test_multilinear_evaluations},
	    {"test_generic_single_inner_product_using_kernel_accumulator", test_inner_product_using_kernel_accumulator},
	    {"test_generic_kernel_add", test_kernel_add},
	    {"test_generic_fri_fold", test_fri_fold},
	    {"test_generic_single_left_right_fold", test_left_right_fold},
	    {"test_extrapolate_line", test_extrapolate_line},
	    {"test_generic_compute_composite", test_compute_composite},
	    {"test_map_kernels", test_map_kernels},
	    {"test_generic_pairwise_product_reduce", test_pairwise_product_reduce},
	    {"generic_test_bivariate_sumcheck_prove_verify", test_bivariate_sumcheck_prove_verify},
	    {"test_additive_ntt", test_ntt},
	    {"test_binary_merkle_vcs_commit_prove_open_correctly", test_binary_merkle_vcs},
	    {"test_fri_commit_fold_query_device_resident", test_fri_commit_fold_query},
	    {"test_prodcheck_layers_and_ring_switch_eq_ind", test_prodcheck_and_ring_switch_callers},
	    {"test_old_hal_prover_state_both_orders_with_switchover", test_old_hal_prover_state},
	};
	int failed = 0;
	try {
		Env env;
		for (auto &t : tests) {
			const int before = g_fail;
			try {
				t.fn(env);
			} catch (const std::exception &ex) {
				std::printf("    exception: %s\n", ex.what());
				g_fail++;
			}
			const bool ok = g_fail == before;
			std::printf("%s %s\n", ok ? "PASS" : "FAIL", t.name);
			if (!ok) failed++;
		}
	} catch (const std::exception &ex) {
		std::printf("FATAL: %s\n", ex.what());
		return 2;
	}
	std::printf("%d/%zu conformance tests passed\n", (int)tests.size() - failed, tests.size());
	return failed ? 1 : 0;
}
