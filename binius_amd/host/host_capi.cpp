// binius_amd/host/host_capi.cpp -> libbinius_amd_host.so
//
// The compiled form of the C++ host mirror (compute_layer.hpp / sumcheck.hpp): a complete
// BivariateSumcheckProver run behind one C call, so the benchmark times the prover loop the way a
// compiled (Rust) host would drive the HAL -- per round: one accumulate_kernels, two scalar
// multiplications, one extrapolate_line per multilinear -- without interpreter overhead between
// HAL calls.  Links only against the C ABI of include/binius_amd.h.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cerrno>
#include <cstring>
#include <thread>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/binius_amd_host.h"
#include <chrono>
#include <cstring>

#include "eq_ind.hpp"
#include "fri.hpp"
#include "piop.hpp"
#include "sumcheck.hpp"

using namespace binius_amd;

namespace {
thread_local std::string g_err;

// ---- RCCL, bound at run time from the librccl.so the process already uses (torch's) -----------
struct rccl_unique_id {
	char internal[128];
};
typedef int (*fn_get_unique_id)(rccl_unique_id *);
typedef int (*fn_comm_init_rank)(void **comm, int nranks, rccl_unique_id id, int rank);
typedef int (*fn_all_gather)(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream);
typedef int (*fn_comm_destroy)(void *comm);
typedef const char *(*fn_get_error_string)(int);
struct rccl_api {
	void *lib = nullptr;
	fn_get_unique_id get_unique_id = nullptr;
	fn_comm_init_rank comm_init_rank = nullptr;
	fn_all_gather all_gather = nullptr;
	fn_comm_destroy comm_destroy = nullptr;
	fn_get_error_string err = nullptr;
} g_rccl;
constexpr int kNcclUint8 = 1; // ncclDataType_t
} // namespace

extern "C" {

const char *bnh_last_error(void) { return g_err.c_str(); }

// ---- RCCL communicator for the sharded prover (one process per GPU) ----------------------------
int bnh_rccl_open(const char *librccl_path)
{
	if (g_rccl.lib) return 0;
	void *h = dlopen(librccl_path, RTLD_NOW | RTLD_GLOBAL);
	if (!h) {
		g_err = std::string("dlopen librccl failed: ") + dlerror();
		return BN_ERR_CORE_LIB;
	}
	g_rccl.lib = h;
	g_rccl.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
	g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
	g_rccl.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
	g_rccl.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
	g_rccl.err = (fn_get_error_string)dlsym(h, "ncclGetErrorString");
	if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_gather || !g_rccl.comm_destroy) {
		g_err = "librccl does not export the expected nccl* symbols";
		return BN_ERR_CORE_LIB;
	}
	return 0;
}
static int rccl_check(int rc, const char *what)
{
	if (rc == 0) return 0;
	g_err = std::string(what) + ": " + (g_rccl.err ? g_rccl.err(rc) : "rccl error");
	return BN_ERR_DEVICE;
}
int bnh_rccl_unique_id(void *out128)
{
	if (!g_rccl.lib) return (g_err = "bnh_rccl_open was not called", BN_ERR_CORE_LIB);
	return rccl_check(g_rccl.get_unique_id((rccl_unique_id *)out128), "ncclGetUniqueId");
}
int bnh_rccl_init(const void *id128, int world, int rank, void **comm_out)
{
	if (!g_rccl.lib) return (g_err = "bnh_rccl_open was not called", BN_ERR_CORE_LIB);
	rccl_unique_id id;
	std::memcpy(&id, id128, sizeof(id));
	return rccl_check(g_rccl.comm_init_rank(comm_out, world, id, rank), "ncclCommInitRank");
}
int bnh_rccl_destroy(void *comm) { return comm && g_rccl.lib ? rccl_check(g_rccl.comm_destroy(comm), "ncclCommDestroy") : 0; }

// Combine hook for the sharded prover: called once per round with the device pointer holding this
// rank's partial (y_1, y_inf) (2 field elements); must return the XOR over all ranks in `evals`.
// ---- intra-node exchange of a few scalars per round through POSIX shared memory ---------------
// The sharded prover needs, every round, the XOR of one 32-byte partial per rank.  All ranks of the
// measured configuration sit on one node, so the cheapest exchange is the host's cache-coherent
// memory: every rank publishes (payload, round) in its own cache line pair and spins on the others'
// (~1 us per round, no launch, no device round trip) -- the device side of a round is then exactly
// the single-GPU one (fused kernel + result mailbox).  The RCCL all_gather path above stays for
// ranks that do not share a node.  Two slots per rank suffice: a rank can only get one round ahead
// (it needs everybody's round r+1 value, published after they have read round r).
struct shm_slot {
	alignas(64) std::atomic<uint64_t> seq;
	uint64_t v[7];
};
struct bnh_shm {
	shm_slot *base = nullptr; // [world][2]
	int world = 0, rank = 0;
	uint64_t round = 0;
	size_t bytes = 0;
	std::string name;
	bool owner = false;
};

int bnh_shm_open(const char *name, int world, int rank, int create, void **out)
{
	if (!name || !out || world < 1 || rank < 0 || rank >= world) return (g_err = "bnh_shm_open: bad arguments", BN_ERR_INPUT_VALIDATION);
	const size_t bytes = sizeof(shm_slot) * 2 * (size_t)world;
	int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
	if (fd < 0) return (g_err = std::string("shm_open failed: ") + strerror(errno), BN_ERR_CORE_LIB);
	if (create && ftruncate(fd, (off_t)bytes) != 0) {
		close(fd);
		shm_unlink(name);
		return (g_err = std::string("ftruncate failed: ") + strerror(errno), BN_ERR_CORE_LIB);
	}
	void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (p == MAP_FAILED) return (g_err = std::string("mmap failed: ") + strerror(errno), BN_ERR_CORE_LIB);
	if (create) std::memset(p, 0, bytes); // seq = 0 everywhere; rounds start at 1
	bnh_shm *h = new bnh_shm;
	h->base = (shm_slot *)p;
	h->world = world;
	h->rank = rank;
	h->bytes = bytes;
	h->name = name;
	h->owner = create != 0;
	*out = h;
	return 0;
}

int bnh_shm_close(void *hv)
{
	bnh_shm *h = (bnh_shm *)hv;
	if (!h) return 0;
	munmap(h->base, h->bytes);
	if (h->owner) shm_unlink(h->name.c_str());
	delete h;
	return 0;
}

// every rank contributes n_words (<= 7) 64-bit words; out[world * n_words] = all of them, rank-major
int bnh_shm_allgather(void *hv, const uint64_t *in, uint32_t n_words, uint64_t *out)
{
	bnh_shm *h = (bnh_shm *)hv;
	if (!h || !in || !out || n_words > 7) return (g_err = "bnh_shm_allgather: bad arguments", BN_ERR_INPUT_VALIDATION);
	const uint64_t r = ++h->round;
	shm_slot &mine = h->base[2 * h->rank + (r & 1)];
	for (uint32_t i = 0; i < n_words; i++) mine.v[i] = in[i];
	mine.seq.store(r, std::memory_order_release);
	for (int w = 0; w < h->world; w++) {
		shm_slot &s = h->base[2 * w + (r & 1)];
		uint64_t spins = 0;
		while (s.seq.load(std::memory_order_acquire) != r) {
			++spins;
			// a rank that is late by more than ~0.1 ms is probably not running (more ranks than granted host
			// CPUs): give the core away between looks instead of burning the quota
			if (spins > (1ull << 14) && (spins & 1023) == 0) std::this_thread::yield();
			if (spins > (1ull << 22)) {
				// slow path: yield every time, and give up after ~20 s (a rank died)
				std::this_thread::yield();
				if (spins > (1ull << 22) + 20000000ull) return (g_err = "shm exchange timed out waiting for a rank", BN_ERR_DEVICE);
			}
		}
		for (uint32_t i = 0; i < n_words; i++) out[(size_t)w * n_words + i] = s.v[i];
	}
	return 0;
}


// One full prove: execute -> fold for n_vars rounds, then finish.
//   d_multilins[m]      device pointers, 2^n_vars elements each (never modified: PreFold)
//   d_scratch           device memory for the folded multilinears, >= m * 2^(n_vars-1) elements
//   comp_indices        2 per composition (IndexComposition<BivariateProduct, 2>)
//   challenges[n_vars]  stand-in for the Fiat-Shamir transcript
//   round_coeffs_out    3 per round ; final_evals_out m
//   reduce              NULL on a single GPU
int bnh_bivariate_sumcheck_prove(bn_ctx *ctx, uint32_t n_vars, uint32_t m, const void *const *d_multilins, void *d_scratch,
                                 uint64_t scratch_elems, uint32_t n_comps, const uint32_t *comp_indices, const bn_f128 *sums,
                                 const bn_f128 *batch_coeff, const bn_f128 *challenges, bn_f128 *round_coeffs_out,
                                 bn_f128 *final_evals_out, bnh_round_reduce_fn reduce, void *reduce_user, void *d_partial,
                                 void *rccl_comm, int world, void *d_gathered, void *shm, int tail_rounds)
{
	// tail_rounds: bit 0 = run the residual rounds inside this call; bit 1 = "peer" exchange: the ranks' partial round
	// evaluations are XORed on the devices, inside the kernels' finalize step (bn_peer_*; the context must be connected),
	// and the host sees the global (y_1, y_inf) in its own mailbox -- `shm` then only carries the one-off rebuild of the
	// residual instance
	const bool peer = (tail_rounds & 2) != 0;
	tail_rounds &= 1;
	struct peer_scope { // local rounds reduced, everything else local
		bn_ctx *c;
		bool on = false;
		void set(bool v)
		{
			if (v != on) {
				if (bn_peer_set_active(c, v ? 1 : 0) != 0) throw Error(Error::DeviceError, bn_last_error());
				on = v;
			}
		}
		~peer_scope()
		{
			if (on) (void)bn_peer_set_active(c, 0);
			(void)bn_host_tail_allow_peer(c, 0);
		}
	} peer_guard{ctx};
	try {
		ComputeLayer hal(ctx);
		DeviceBumpAllocator dev_alloc(FSliceMut{d_scratch, (size_t)scratch_elems});
		std::vector<B128> host_mem(m + 4);
		HostBumpAllocator host_alloc(HostSliceMut{host_mem.data(), host_mem.size()});
		std::vector<FSlice> mls;
		for (uint32_t j = 0; j < m; j++) mls.push_back(FSlice{d_multilins[j], (size_t)1 << n_vars});
		std::vector<IndexCompositionBivariate> comps;
		std::vector<B128> sv;
		for (uint32_t c = 0; c < n_comps; c++) {
			comps.push_back(IndexCompositionBivariate{m, {comp_indices[2 * c], comp_indices[2 * c + 1]}});
			sv.emplace_back(sums[c].lo, sums[c].hi);
		}
		const B128 bc(batch_coeff->lo, batch_coeff->hi);
		if (!reduce && !rccl_comm && !shm && !peer) {
			const auto t_prof = std::chrono::steady_clock::now();
			if (AbiProf::on()) AbiProf::get() = AbiProf{};
			BivariateSumcheckProver prover(hal, dev_alloc, host_alloc, n_vars, comps, sv, mls);
			for (uint32_t r = 0; r < n_vars; r++) {
				std::vector<B128> rc = prover.execute(bc);
				for (size_t i = 0; i < 3 && i < rc.size(); i++) round_coeffs_out[3 * r + i] = rc[i].raw();
				prover.fold(B128(challenges[r].lo, challenges[r].hi));
			}
			std::vector<B128> fin = prover.finish();
			for (uint32_t j = 0; j < m; j++) final_evals_out[j] = fin[j].raw();
			if (AbiProf::on()) { // BNH_PROF=1: where the wall time of this prove went (us): inside the C ABI by kind of call, and the rest
				const AbiProf &p = AbiProf::get();
				const double tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_prof).count();
				fprintf(stderr, "[bnh prof] n_vars %u: total %.2f us; kernel_launch %.2f (%llu calls), extrapolate_line %.2f (%llu), copies %.2f (%llu); outside the ABI %.2f\n",
				        n_vars, tot, p.ns[0] / 1e3, (unsigned long long)p.calls[0], p.ns[1] / 1e3, (unsigned long long)p.calls[1], p.ns[2] / 1e3,
				        (unsigned long long)p.calls[2], tot - (p.ns[0] + p.ns[1] + p.ns[2]) / 1e3);
			}
			return 0;
		}
		// sharded variant: same state machine, round evals combined across ranks by `reduce`
		std::vector<ExprEval> evaluators;
		for (const auto &c : comps) evaluators.push_back(hal.compile_expr(c.expression()));
		std::vector<FSliceMut> cur;
		for (const auto &ml : mls) cur.push_back(FSliceMut{const_cast<void *>(ml.ptr), ml.len_});
		bool pre_fold = true;
		B128 running = evaluate_univariate(sv, bc);
		const std::vector<B128> coeffs = powers(bc, n_comps);
		// `n_rounds` rounds on the arrays in `cur` (2^n_rounds elements each); exchange: combine the
		// ranks' partial round evaluations (local rounds) or not (residual rounds, identical everywhere)
		// peer exchange + host tail: from the launch after which the library holds the arrays on the host (bn_host_tail_active)
		// the rounds' LOCAL partial sums come back from host arithmetic and meet in the shared-memory segment instead
		bool host_rounds = false;
		auto do_rounds = [&](uint32_t n_rounds, bool exchange, const bn_f128 *ch, bn_f128 *coeffs_out) {
		for (uint32_t r = 0; r < n_rounds; r++) {
			const size_t rem = n_rounds - r, split = rem - 1;
			std::vector<KernelMemMap> maps;
			for (auto &ml : cur) {
				auto h = ComputeMemory::split_half(ComputeMemory::as_const(ml));
				maps.push_back(KernelMemMap::chunked(h.first, 0));
				maps.push_back(KernelMemMap::chunked(h.second, 0));
				maps.push_back(KernelMemMap::local(split));
			}
			bn_f128 ev[2];
			if (shm || !exchange) {
				// device side identical to the single-GPU round (result through the mailbox); the
				// partials of the ranks are combined in host shared memory
				std::vector<FSlice> cmls;
				for (auto &ml : cur) cmls.push_back(ComputeMemory::as_const(ml));
				const std::vector<B128> part = calculate_round_evals(hal, rem, bc, cmls, evaluators);
				const uint64_t mine[4] = {part[0].raw().lo, part[0].raw().hi, part[1].raw().lo, part[1].raw().hi};
				std::vector<uint64_t> all((size_t)4 * world);
				const bool via_shm = exchange && (!peer || host_rounds);
				const int n_src = via_shm ? world : 1;
				if (via_shm) {
					if (bnh_shm_allgather(shm, mine, 4, all.data())) throw Error(Error::DeviceError, g_err);
				} else {
					for (int i = 0; i < 4; i++) all[i] = mine[i];
				}
				if (exchange && peer && !host_rounds) {
					int active = 0;
					check(bn_host_tail_active(ctx, &active));
					if (active) { // (every rank sees this after the same launch: the shards have the same size)
						host_rounds = true;
						peer_guard.set(false);
					}
				}
				ev[0] = bn_f128{0, 0};
				ev[1] = bn_f128{0, 0};
				for (int w = 0; w < n_src; w++) {
					ev[0].lo ^= all[4 * w + 0];
					ev[0].hi ^= all[4 * w + 1];
					ev[1].lo ^= all[4 * w + 2];
					ev[1].hi ^= all[4 * w + 3];
				}
			} else {
				hal.execute([&](ComputeLayerExecutor &exec) {
					exec.accumulate_kernels_to_device(
					    [&](KernelExecutor &ke, size_t log_chunks, std::vector<KernelBuffer> &b) {
						    const size_t lcs = split - log_chunks;
						    KernelValue a1 = ke.decl_value(B128::ZERO());
						    std::vector<KSlice> rows;
						    for (uint32_t i = 0; i < m; i++) rows.push_back(b[3 * i + 1].to_ref());
						    SlicesBatch<KSlice> e1(rows, (size_t)1 << lcs);
						    for (uint32_t c = 0; c < n_comps; c++) ke.sum_composition_evals(e1, evaluators[c], coeffs[c], a1);
						    for (uint32_t i = 0; i < m; i++) ke.add(lcs, b[3 * i].to_ref(), b[3 * i + 1].to_ref(), b[3 * i + 2].as_mut());
						    KernelValue ai = ke.decl_value(B128::ZERO());
						    rows.clear();
						    for (uint32_t i = 0; i < m; i++) rows.push_back(b[3 * i + 2].to_ref());
						    SlicesBatch<KSlice> ei(rows, (size_t)1 << lcs);
						    for (uint32_t c = 0; c < n_comps; c++) ke.sum_composition_evals(ei, evaluators[c], coeffs[c], ai);
						    return std::vector<KernelValue>{a1, ai};
					    },
					    maps, d_partial);
					return std::vector<B128>{};
				});
				if (rccl_comm) {
					// ONE RCCL collective per round: all_gather of this rank's 32-byte partial on the
					// context's stream (stream-ordered after the kernels), then XOR of the G partials
					void *stream = nullptr;
					check(bn_ctx_get_stream(ctx, &stream));
					if (g_rccl.all_gather(d_partial, d_gathered, 32, kNcclUint8, rccl_comm, stream) != 0)
						throw Error(Error::DeviceError, "ncclAllGather failed");
					// XOR of the G partials on the device, result through the zero-copy mailbox
					check(bn_xor_reduce(ctx, d_gathered, (uint32_t)world, 2, ev));
				} else if (reduce(reduce_user, d_partial, ev)) {
					throw Error(Error::CoreLibError, "round reduce callback failed");
				}
			}
			std::vector<B128> rc = calculate_round_coeffs_from_evals(running, {B128(ev[0].lo, ev[0].hi), B128(ev[1].lo, ev[1].hi)});
			for (size_t i = 0; i < 3; i++) coeffs_out[3 * r + i] = rc[i].raw();
			const B128 z(ch[r].lo, ch[r].hi);
			running = evaluate_univariate(rc, z);
			struct FoldArgs {
				FSliceMut evals_0;
				FSlice evals_1;
			};
			std::vector<FoldArgs> prepared;
			for (auto &ml : cur) {
				auto h = ComputeMemory::split_half_mut(ml);
				FSliceMut e0 = h.first;
				if (pre_fold) {
					e0 = dev_alloc.alloc(h.first.len_);
					hal.copy_d2d(ComputeMemory::as_const(h.first), e0);
				}
				prepared.push_back(FoldArgs{e0, ComputeMemory::as_const(h.second)});
				ml = e0;
			}
			// one `map` scope over the multilinears (v3/bivariate_product.rs:217-228): one fold batch,
			// which the ABI runs together with the next round's evaluation
			hal.execute([&](ComputeLayerExecutor &exec) {
				exec.map(prepared.begin(), prepared.end(), [&](ComputeLayerExecutor &e, FoldArgs &a) {
					e.extrapolate_line(a.evals_0, a.evals_1, z);
					return 0;
				});
				return std::vector<B128>{};
			});
			pre_fold = false;
		}
		};
		if (peer) {
			if (!shm) throw Error(Error::InputValidation, "the peer exchange needs the shared-memory segment for the residual instance");
			check(bn_host_tail_allow_peer(ctx, 1)); // (the host rounds' partials are exchanged here, through the segment)
			peer_guard.set(true);
		}
		do_rounds(n_vars, true, challenges, round_coeffs_out);
		peer_guard.set(false);
		uint32_t log_world = 0;
		while ((1 << (log_world + 1)) <= world) log_world++;
		if (rccl_comm && !shm && tail_rounds && log_world > 0) {
			// ---- residual rounds, RCCL transport: every rank is down to one element per multilinear on the
			// device.  One ncclAllGather per multilinear (16 bytes per rank, on the context's stream) rebuilds
			// the residual multilinears of `world` elements (index = rank) in device memory; the last
			// log2(world) rounds run on them, identically on every rank, with no further exchange.
			void *stream = nullptr;
			check(bn_ctx_get_stream(ctx, &stream)); // flushes the deferred last fold
			std::vector<FSliceMut> res;
			for (uint32_t j = 0; j < m; j++) res.push_back(dev_alloc.alloc((size_t)world));
			for (uint32_t j = 0; j < m; j++)
				if (g_rccl.all_gather(cur[j].ptr, res[j].ptr, 16, kNcclUint8, rccl_comm, stream) != 0)
					throw Error(Error::DeviceError, "ncclAllGather (residual multilinears) failed");
			for (uint32_t j = 0; j < m; j++) cur[j] = res[j];
			pre_fold = true; // fold into fresh scratch, as in the first local round
			do_rounds(log_world, false, challenges + n_vars, round_coeffs_out + 3 * n_vars);
		} else if (shm && tail_rounds && log_world > 0) {
			// ---- residual rounds: every rank is down to one element per multilinear.  One exchange of
			// the m local finals rebuilds the m residual multilinears of `world` elements (index = rank)
			// in pinned, device-visible host memory (no upload); the last log2(world) rounds run on them.
			std::vector<B128> fin(m);
			for (uint32_t j = 0; j < m; j++) hal.copy_d2h(ComputeMemory::as_const(cur[j]), &fin[j], 1);
			void *h_scr = nullptr, *d_scr = nullptr;
			uint64_t scr_elems = 0;
			check(bn_host_scratch(ctx, &h_scr, &d_scr, &scr_elems));
			if ((uint64_t)m * world > scr_elems) throw Error(Error::InputValidation, "residual instance does not fit the pinned scratch");
			bn_f128 *res = (bn_f128 *)h_scr;
			for (uint32_t j0 = 0; j0 < m; j0 += 3) {
				const uint32_t nj = (m - j0) < 3 ? (m - j0) : 3;
				uint64_t mine[6];
				for (uint32_t j = 0; j < nj; j++) {
					mine[2 * j] = fin[j0 + j].raw().lo;
					mine[2 * j + 1] = fin[j0 + j].raw().hi;
				}
				std::vector<uint64_t> all((size_t)2 * nj * world);
				if (bnh_shm_allgather(shm, mine, 2 * nj, all.data())) throw Error(Error::DeviceError, g_err);
				for (int w = 0; w < world; w++)
					for (uint32_t j = 0; j < nj; j++) res[(size_t)(j0 + j) * world + w] = bn_f128{all[(size_t)w * 2 * nj + 2 * j], all[(size_t)w * 2 * nj + 2 * j + 1]};
			}
			std::atomic_thread_fence(std::memory_order_seq_cst);
			for (uint32_t j = 0; j < m; j++) cur[j] = FSliceMut{(char *)d_scr + (size_t)j * world * sizeof(bn_f128), (size_t)world};
			pre_fold = true; // the pinned inputs are read-only: fold into device scratch
			do_rounds(log_world, false, challenges + n_vars, round_coeffs_out + 3 * n_vars);
		}
		for (uint32_t j = 0; j < m; j++) {
			B128 v;
			hal.copy_d2h(ComputeMemory::as_const(cur[j]), &v, 1);
			final_evals_out[j] = v.raw();
		}
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

static thread_local int g_mlecheck_mode = -1;
int bnh_mlecheck_last_mode(void) { return g_mlecheck_mode; }

// ---- MLE-check prover behind a handle: SumcheckProver::{execute, fold, finish} (prove/batch_sumcheck.rs:38-70) one
// call each, so that a host whose challenges come out of a transcript can drive it round by round.
//   d_eq_ind             2^(n_vars-1) elements: tensor expansion of eq_ind_challenges[0 .. n_vars-1)
struct bnh_mlecheck {
	ComputeLayer hal;
	DeviceBumpAllocator dev_alloc;
	std::vector<B128> host_mem;
	HostBumpAllocator host_alloc;
	std::unique_ptr<WeightedMLEcheckProver> weighted;
	std::unique_ptr<BivariateMLEcheckProver> literal;
	uint32_t m;
	bnh_mlecheck(bn_ctx *ctx, void *d_scratch, uint64_t scratch_elems, uint32_t m_)
	    : hal(ctx), dev_alloc(FSliceMut{d_scratch, (size_t)scratch_elems}), host_mem(m_ + 4),
	      host_alloc(HostSliceMut{host_mem.data(), host_mem.size()}), m(m_)
	{
	}
};

int bnh_mlecheck_new(bn_ctx *ctx, uint32_t n_vars, uint32_t m, const void *const *d_multilins, const void *d_eq_ind,
                     const bn_f128 *eq_ind_challenges, void *d_scratch, uint64_t scratch_elems, uint32_t n_comps,
                     const uint32_t *comp_indices, const bn_f128 *sums, bnh_mlecheck **out)
{
	try {
		if (!out) throw Error(Error::InputValidation, "null argument");
		std::unique_ptr<bnh_mlecheck> h(new bnh_mlecheck(ctx, d_scratch, scratch_elems, m));
		std::vector<FSlice> mls;
		for (uint32_t j = 0; j < m; j++) mls.push_back(FSlice{d_multilins[j], (size_t)1 << n_vars});
		std::vector<IndexCompositionBivariate> comps;
		std::vector<B128> sv, eqc;
		for (uint32_t c = 0; c < n_comps; c++) {
			comps.push_back(IndexCompositionBivariate{m, {comp_indices[2 * c], comp_indices[2 * c + 1]}});
			sv.emplace_back(sums[c].lo, sums[c].hi);
		}
		for (uint32_t i = 0; i < n_vars; i++) eqc.emplace_back(eq_ind_challenges[i].lo, eq_ind_challenges[i].hi);
		const FSlice eq_table{d_eq_ind, (size_t)1 << (n_vars ? n_vars - 1 : 0)};
		// The weighted prover (sumcheck.hpp) when it applies: a proper 2-colouring of the compositions, invertible
		// indicator coordinates, enough scratch, and a table that IS the tensor expansion of the coordinates (the
		// constructor's contract: "an existing tensor expansion for eq_ind_challenges", bivariate_mlecheck.rs:69-71 --
		// spot-checked here on n_vars + 17 entries because the weighted prover derives the later tables from the
		// coordinates instead of folding the given one).
		std::vector<bool> weighted;
		const char *mode = getenv("BN_MLECHECK");
		if (!(mode && std::string(mode) == "eager") && n_vars >= 2 && n_comps > 0) {
			weighted = WeightedMLEcheckProver::colouring(m, comps);
			bool any = false;
			for (bool w : weighted) any = any || w;
			if (!any || !WeightedMLEcheckProver::coordinates_invertible(eqc, n_vars) ||
			    WeightedMLEcheckProver::required_device_memory(weighted, n_vars) > scratch_elems)
				weighted.clear();
		}
		if (!weighted.empty()) {
			std::vector<uint64_t> offs{0, eq_table.len() - 1};
			for (uint32_t i = 0; i + 1 < n_vars; i++) offs.push_back((uint64_t)1 << i);
			uint64_t lcg = eqc[0].raw().lo ^ 0x9E3779B97F4A7C15ull; // a few more, spread over the table
			for (int k = 0; k < 16; k++) {
				lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
				offs.push_back((lcg >> 20) & (eq_table.len() - 1));
			}
			std::vector<B128> got(offs.size());
			check(bn_gather_d2h(ctx, d_eq_ind, offs.data(), offs.size(), 1, reinterpret_cast<bn_f128 *>(got.data())));
			for (size_t k = 0; k < offs.size() && !weighted.empty(); k++) {
				B128 want = B128::ONE();
				for (uint32_t i = 0; i + 1 < n_vars; i++) want = want * (((offs[k] >> i) & 1) ? eqc[i] : B128::ONE() - eqc[i]);
				if (!(got[k] == want)) weighted.clear();
			}
		}
		g_mlecheck_mode = weighted.empty() ? 0 : 1;
		if (!weighted.empty())
			h->weighted.reset(new WeightedMLEcheckProver(h->hal, h->dev_alloc, h->host_alloc, n_vars, comps, sv, mls, eq_table, eqc, weighted));
		else
			h->literal.reset(new BivariateMLEcheckProver(h->hal, h->dev_alloc, h->host_alloc, n_vars, comps, sv, mls, eq_table, eqc));
		*out = h.release();
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

// coeffs_out[4]: the round polynomial (degree 3)
int bnh_mlecheck_execute(bnh_mlecheck *h, const bn_f128 *batch_coeff, bn_f128 *coeffs_out)
{
	try {
		if (!h || !batch_coeff || !coeffs_out) throw Error(Error::InputValidation, "null argument");
		const B128 bc(batch_coeff->lo, batch_coeff->hi);
		const std::vector<B128> rc = h->weighted ? h->weighted->execute(bc) : h->literal->execute(bc);
		for (size_t i = 0; i < 4; i++) coeffs_out[i] = i < rc.size() ? rc[i].raw() : bn_f128{0, 0};
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

int bnh_mlecheck_fold(bnh_mlecheck *h, const bn_f128 *challenge)
{
	try {
		if (!h || !challenge) throw Error(Error::InputValidation, "null argument");
		const B128 z(challenge->lo, challenge->hi);
		if (h->weighted)
			h->weighted->fold(z);
		else
			h->literal->fold(z);
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

// final_evals_out[m + 1]: the multilinears' evaluations, then eq_ind_prefix_eval.  The handle stays valid (free it).
int bnh_mlecheck_finish(bnh_mlecheck *h, bn_f128 *final_evals_out)
{
	try {
		if (!h || !final_evals_out) throw Error(Error::InputValidation, "null argument");
		const std::vector<B128> fin = h->weighted ? h->weighted->finish() : h->literal->finish();
		for (uint32_t j = 0; j <= h->m; j++) final_evals_out[j] = fin[j].raw();
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

void bnh_mlecheck_free(bnh_mlecheck *h) { delete h; }

// One complete BivariateMLEcheckProver run (v3/bivariate_mlecheck.rs) behind a C call.
//   round_coeffs_out     [4 * n_vars] (degree-3 round polynomials)
//   final_evals_out      [m + 1]      (the last one is eq_ind_prefix_eval)
int bnh_bivariate_mlecheck_prove(bn_ctx *ctx, uint32_t n_vars, uint32_t m, const void *const *d_multilins, const void *d_eq_ind,
                                 const bn_f128 *eq_ind_challenges, void *d_scratch, uint64_t scratch_elems, uint32_t n_comps,
                                 const uint32_t *comp_indices, const bn_f128 *sums, const bn_f128 *batch_coeff,
                                 const bn_f128 *challenges, bn_f128 *round_coeffs_out, bn_f128 *final_evals_out)
{
	bnh_mlecheck *h = nullptr;
	int rc = bnh_mlecheck_new(ctx, n_vars, m, d_multilins, d_eq_ind, eq_ind_challenges, d_scratch, scratch_elems, n_comps, comp_indices, sums, &h);
	for (uint32_t r = 0; rc == 0 && r < n_vars; r++) {
		rc = bnh_mlecheck_execute(h, batch_coeff, round_coeffs_out + 4 * r);
		if (rc == 0) rc = bnh_mlecheck_fold(h, &challenges[r]);
	}
	if (rc == 0) rc = bnh_mlecheck_finish(h, final_evals_out);
	bnh_mlecheck_free(h);
	return rc;
}

// FRI commit phase + all fold rounds + finalize through the C++ mirror (fri.hpp), everything on the device.
//   d_message        2^(log_dim + log_batch_size) elements (the interleaved message)
//   d_scratch        device memory for the codeword, the folded codewords and the Merkle trees:
//                    2 * 2^(log_dim + log_batch_size + log_inv_rate) elements are always enough
//   challenges       [log_dim + log_batch_size]
//   roots_out        [(n_arities + 1) * 32 bytes]: the commitment, then one root per committed oracle
//   terminate_out    [2^(log_inv_rate + n_final_challenges)] elements or NULL
//   phase_ms_out     [2] wall-clock milliseconds of the commit phase and of the fold phase, or NULL
int bnh_fri_commit_fold(bn_ctx *ctx, uint32_t log_dim, uint32_t log_inv_rate, uint32_t log_batch_size, const uint32_t *fold_arities,
                        uint32_t n_arities, uint32_t n_test_queries, const void *d_message, void *d_scratch, uint64_t scratch_elems,
                        const bn_f128 *challenges, uint8_t *roots_out, bn_f128 *terminate_out, double *phase_ms_out)
{
	try {
		ComputeLayer hal(ctx);
		DeviceBumpAllocator dev_alloc(FSliceMut{d_scratch, (size_t)scratch_elems});
		FRIParams p(log_dim, log_inv_rate, log_batch_size, std::vector<size_t>(fold_arities, fold_arities + n_arities), n_test_queries);
		AdditiveNTT ntt(hal, 5, p.rs_log_len());
		BinaryMerkleTreeProver merkle(hal);
		hal.sync();
		const auto t0 = std::chrono::steady_clock::now();
		CommitOutput out = commit_interleaved(hal, dev_alloc, p, ntt, merkle, FSlice{d_message, (size_t)1 << (log_dim + log_batch_size)});
		hal.sync();
		const auto t1 = std::chrono::steady_clock::now();
		std::memcpy(roots_out, out.commitment.data(), 32);
		FRIFolder folder(hal, p, ntt, merkle, ComputeMemory::as_const(out.codeword), out.committed);
		size_t n_roots = 1;
		for (size_t r = 0; r < folder.n_rounds(); r++) {
			auto [has_root, root] = folder.execute_fold_round(dev_alloc, B128(challenges[r].lo, challenges[r].hi));
			if (has_root) std::memcpy(roots_out + 32 * n_roots++, root.data(), 32);
		}
		auto fin = folder.finalize();
		hal.sync();
		const auto t2 = std::chrono::steady_clock::now();
		if (terminate_out)
			for (size_t i = 0; i < fin.first.size(); i++) terminate_out[i] = fin.first[i].raw();
		if (phase_ms_out) {
			phase_ms_out[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
			phase_ms_out[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
		}
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

// piop::prove (crates/core/src/piop/prove.rs:148-395) through the C++ mirror piop.hpp: commit_interleaved of the merged message,
// one BivariateSumcheckProver per number of variables, the front-loaded batch prover interleaved with the FRI folder.
// The transcript comes back as a list of items in writing order: items_out[2 i] = kind (0 round proof, 1 final evaluations of a
// finished prover, 2 FRI round commitment, 3 FRI terminate codeword), items_out[2 i + 1] = number of scalars (kinds 0, 1, 3:
// consumed from scalars_out in order) or 1 digest (kind 2: consumed from digests_out in order).
int bnh_piop_prove(bn_ctx *ctx, uint32_t n_committed, const uint32_t *committed_n_vars, const void *const *d_committed, uint32_t n_transparent,
                   const uint32_t *transparent_n_vars, const void *const *d_transparent, uint32_t n_claims, const uint32_t *claims, const bn_f128 *claim_sums,
                   uint32_t log_dim, uint32_t log_inv_rate, uint32_t log_batch_size, const uint32_t *fold_arities, uint32_t n_arities, uint32_t n_test_queries,
                   const void *d_message, void *d_scratch, uint64_t scratch_elems, const bn_f128 *batch_coeffs, uint32_t n_batch_coeffs,
                   const bn_f128 *challenges, uint32_t n_challenges, uint8_t *commitment_out, uint32_t *items_out, uint32_t max_items, uint32_t *n_items_out,
                   bn_f128 *scalars_out, uint64_t max_scalars, uint64_t *n_scalars_out, uint8_t *digests_out, uint32_t max_digests, uint32_t *n_digests_out,
                   double *phase_ms_out)
{
	try {
		if (!ctx || !commitment_out || !items_out || !n_items_out || !scalars_out || !n_scalars_out || !digests_out || !n_digests_out)
			throw Error(Error::InputValidation, "null argument");
		if ((n_committed && (!committed_n_vars || !d_committed)) || (n_transparent && (!transparent_n_vars || !d_transparent)) || (n_claims && (!claims || !claim_sums)) ||
		    (n_arities && !fold_arities) || !d_message || (n_batch_coeffs && !batch_coeffs) || (n_challenges && !challenges))
			throw Error(Error::InputValidation, "null argument");
		for (uint32_t i = 0; i < n_committed; i++)
			if (committed_n_vars[i] >= 48 || !d_committed[i]) throw Error(Error::InputValidation, "committed multilinear: null, or its number of variables out of range (< 48)");
		for (uint32_t i = 0; i < n_transparent; i++)
			if (transparent_n_vars[i] >= 48 || !d_transparent[i]) throw Error(Error::InputValidation, "transparent multilinear: null, or its number of variables out of range (< 48)");
		if (log_dim + log_batch_size + log_inv_rate >= 48) throw Error(Error::InputValidation, "FRI parameters out of range");
		ComputeLayer hal(ctx);
		DeviceBumpAllocator dev_alloc(FSliceMut{d_scratch, (size_t)scratch_elems});
		std::vector<size_t> n_varss;
		std::vector<FSlice> committed, transparents;
		size_t host_elems = 8;
		for (uint32_t i = 0; i < n_committed; i++) {
			if (i && committed_n_vars[i] < committed_n_vars[i - 1]) throw PiopError("CommittedsNotSorted");
			n_varss.push_back(committed_n_vars[i]);
			committed.push_back(FSlice{d_committed[i], (size_t)1 << committed_n_vars[i]});
			host_elems++;
		}
		for (uint32_t i = 0; i < n_transparent; i++) {
			transparents.push_back(FSlice{d_transparent[i], (size_t)1 << transparent_n_vars[i]});
			host_elems++;
		}
		std::vector<B128> host_mem(host_elems);
		HostBumpAllocator host_alloc(HostSliceMut{host_mem.data(), host_mem.size()});
		const CommitMeta commit_meta = CommitMeta::with_vars(n_varss);
		if (commit_meta.total_vars() != (size_t)log_dim + log_batch_size) throw PiopError("FRI message length does not match the commit metadata's total_vars");
		std::vector<PIOPSumcheckClaim> cl;
		for (uint32_t i = 0; i < n_claims; i++)
			cl.push_back(PIOPSumcheckClaim{claims[3 * i], claims[3 * i + 1], claims[3 * i + 2], B128(claim_sums[i].lo, claim_sums[i].hi)});
		std::vector<B128> bcs, chs;
		for (uint32_t i = 0; i < n_batch_coeffs; i++) bcs.emplace_back(batch_coeffs[i].lo, batch_coeffs[i].hi);
		for (uint32_t i = 0; i < n_challenges; i++) chs.emplace_back(challenges[i].lo, challenges[i].hi);
		FRIParams p(log_dim, log_inv_rate, log_batch_size, std::vector<size_t>(fold_arities, fold_arities + n_arities), n_test_queries);
		AdditiveNTT ntt(hal, 5, p.rs_log_len());
		BinaryMerkleTreeProver merkle(hal);
		hal.sync();
		const auto t0 = std::chrono::steady_clock::now();
		CommitOutput co = commit_interleaved(hal, dev_alloc, p, ntt, merkle, FSlice{d_message, (size_t)1 << (log_dim + log_batch_size)});
		hal.sync();
		const auto t1 = std::chrono::steady_clock::now();
		std::memcpy(commitment_out, co.commitment.data(), 32);
		if (AbiProf::on()) AbiProf::get() = AbiProf{};
		PiopProveOutput out = piop_prove(hal, dev_alloc, host_alloc, p, ntt, merkle, commit_meta, co.committed, ComputeMemory::as_const(co.codeword), committed,
		                                 transparents, cl, bcs, chs);
		hal.sync();
		const auto t2 = std::chrono::steady_clock::now();
		if (AbiProf::on()) {
			const AbiProf &pr = AbiProf::get();
			fprintf(stderr, "[bnh prof] piop prove %.1f us: send_round_proof %.1f, receive_challenge %.1f, fri round %.1f | inside the ABI: kernel_launch %.1f (%llu calls), extrapolate_line %.1f (%llu), copies %.1f (%llu)\n",
			        std::chrono::duration<double, std::micro>(t2 - t1).count(), out.phase_ns[0] / 1e3, out.phase_ns[1] / 1e3, out.phase_ns[2] / 1e3, pr.ns[0] / 1e3,
			        (unsigned long long)pr.calls[0], pr.ns[1] / 1e3, (unsigned long long)pr.calls[1], pr.ns[2] / 1e3, (unsigned long long)pr.calls[2]);
		}
		uint32_t ni = 0, nd = 0;
		uint64_t ns = 0;
		for (const auto &it : out.transcript.items) {
			if (ni >= max_items) throw Error(Error::InputValidation, "transcript has more items than items_out holds");
			items_out[2 * ni] = (uint32_t)it.kind;
			if (it.kind == PiopTranscript::Item::FriCommitment) {
				if (nd >= max_digests) throw Error(Error::InputValidation, "transcript has more digests than digests_out holds");
				std::memcpy(digests_out + 32 * (size_t)nd++, it.digest.data(), 32);
				items_out[2 * ni + 1] = 1;
			} else {
				if (ns + it.scalars.size() > max_scalars) throw Error(Error::InputValidation, "transcript has more scalars than scalars_out holds");
				for (const B128 &v : it.scalars) scalars_out[ns++] = v.raw();
				items_out[2 * ni + 1] = (uint32_t)it.scalars.size();
			}
			ni++;
		}
		*n_items_out = ni;
		*n_scalars_out = ns;
		*n_digests_out = nd;
		if (phase_ms_out) {
			phase_ms_out[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
			phase_ms_out[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
		}
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

// The front-loaded batch prover alone (protocols/sumcheck/prove/front_loaded.rs:33-203: BatchProver::run with the transcript's
// samples handed in): p BivariateSumcheckProvers on ONE layer, ascending by number of variables.
//   prover_desc[3 * i] = (n_vars, m, n_comps) of prover i; d_multilins: the m pointers of prover 0, then of prover 1, ...;
//   comp_indices / sums: likewise concatenated; batch_coeffs[n_provers]; challenges[max n_vars]
//   round_proofs_out[2 * total_rounds]: the truncated round polynomial of every round (missing coefficients zero);
//   final_evals_out: the provers' final evaluations concatenated in finishing (= input) order
int bnh_batch_sumcheck_prove(bn_ctx *ctx, uint32_t n_provers, const uint32_t *prover_desc, const void *const *d_multilins, const uint32_t *comp_indices,
                             const bn_f128 *sums, void *d_scratch, uint64_t scratch_elems, const bn_f128 *batch_coeffs, const bn_f128 *challenges,
                             bn_f128 *round_proofs_out, bn_f128 *final_evals_out)
{
	try {
		// (public entry point: every pointer it dereferences and every shift amount it derives is checked first -- ADVICE r5)
		if (!ctx || (n_provers && (!prover_desc || !batch_coeffs)) || !challenges || !round_proofs_out || !final_evals_out || (!d_scratch && scratch_elems))
			throw Error(Error::InputValidation, "null argument");
		size_t total_m = 0, total_c = 0;
		for (uint32_t i = 0; i < n_provers; i++) {
			if (prover_desc[3 * i] >= 48) throw Error(Error::InputValidation, "a prover's number of variables is out of range (< 48)");
			if (i && prover_desc[3 * i] < prover_desc[3 * (i - 1)]) throw Error(Error::InputValidation, "ClaimsOutOfOrder: provers ascend by number of variables");
			total_m += prover_desc[3 * i + 1];
			total_c += prover_desc[3 * i + 2];
		}
		if ((total_m && !d_multilins) || (total_c && (!comp_indices || !sums))) throw Error(Error::InputValidation, "null argument");
		for (size_t j = 0; j < total_m; j++)
			if (!d_multilins[j]) throw Error(Error::InputValidation, "null multilinear");
		ComputeLayer hal(ctx);
		DeviceBumpAllocator dev_alloc(FSliceMut{d_scratch, (size_t)scratch_elems});
		std::vector<B128> host_mem(total_m + 8);
		HostBumpAllocator host_alloc(HostSliceMut{host_mem.data(), host_mem.size()});
		std::vector<std::unique_ptr<BivariateSumcheckProver>> provers;
		std::vector<B128> bcs;
		size_t at_ml = 0, at_c = 0;
		for (uint32_t i = 0; i < n_provers; i++) {
			const uint32_t n_vars = prover_desc[3 * i], m = prover_desc[3 * i + 1], nc = prover_desc[3 * i + 2];
			std::vector<FSlice> mls;
			for (uint32_t j = 0; j < m; j++) mls.push_back(FSlice{d_multilins[at_ml + j], (size_t)1 << n_vars});
			std::vector<IndexCompositionBivariate> comps;
			std::vector<B128> sv;
			for (uint32_t c = 0; c < nc; c++) {
				comps.push_back(IndexCompositionBivariate{m, {comp_indices[2 * (at_c + c)], comp_indices[2 * (at_c + c) + 1]}});
				sv.emplace_back(sums[at_c + c].lo, sums[at_c + c].hi);
			}
			at_ml += m;
			at_c += nc;
			provers.push_back(std::make_unique<BivariateSumcheckProver>(hal, dev_alloc, host_alloc, n_vars, comps, sv, mls));
			bcs.emplace_back(batch_coeffs[i].lo, batch_coeffs[i].hi);
		}
		SumcheckBatchProver batch(std::move(provers), bcs);
		const size_t rounds = batch.total_rounds();
		PiopTranscript tr;
		for (size_t r = 0; r < rounds; r++) {
			batch.send_round_proof(tr);
			batch.receive_challenge(B128(challenges[r].lo, challenges[r].hi));
		}
		batch.finish(tr);
		size_t r = 0, fe = 0;
		for (const auto &it : tr.items) {
			if (it.kind == PiopTranscript::Item::RoundProof) {
				for (size_t i = 0; i < 2; i++) round_proofs_out[2 * r + i] = i < it.scalars.size() ? it.scalars[i].raw() : bn_f128{0, 0};
				r++;
			} else if (it.kind == PiopTranscript::Item::MultilinearEvals) {
				for (const B128 &v : it.scalars) final_evals_out[fe++] = v.raw();
			}
		}
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

// EqIndSumcheckProver (crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644; binius_amd/host/eq_ind.hpp) over the old HAL
// (binius_hal::ComputationBackend -> bn_hal_round_evals / the ComputeLayer's folds), High-to-Low, compositions of degree 2: the
// zerocheck of a constraint set -- one composition per constraint over ALL multilinears of the table
// (core/src/constraint_system/prove.rs:431-505).
//   d_multilins[n_mls]: 2^n_vars elements each, FOLDED IN PLACE;  steps / steps_inf: the n_comps compositions and their leading
//   forms, concatenated (n_steps[c] / n_steps_inf[c] steps each);  sums[n_comps];  eq_ind_challenges[n_vars];
//   d_eq_ind: 2^(n_vars - 1) elements of scratch for the indicator's partial evaluations (expanded here, eq_ind.rs:430-446)
//   round_coeffs_out[(D + 2) * n_vars], D = max(2, largest degree): the round polynomials have degree D + 1 (D = 2 for the tables'
//   constraints: four coefficients per round); final_evals_out[n_mls + 1] (the last one: the indicator's prefix evaluation)
int bnh_eqind_sumcheck_prove(bn_ctx *ctx, uint32_t n_vars, uint32_t n_mls, void *const *d_multilins, uint32_t n_comps, const bn_step *steps,
                             const uint32_t *n_steps, const bn_step *steps_inf, const uint32_t *n_steps_inf, const uint32_t *degrees, const bn_f128 *sums,
                             const bn_f128 *eq_ind_challenges, void *d_eq_ind, uint64_t eq_ind_elems, const bn_f128 *batch_coeff, const bn_f128 *challenges,
                             bn_f128 *round_coeffs_out, bn_f128 *final_evals_out)
{
	try {
		if (!ctx || (!d_multilins && n_mls) || (n_comps && (!steps || !n_steps || !steps_inf || !n_steps_inf || !sums)) || !eq_ind_challenges || !d_eq_ind || !batch_coeff ||
		    !challenges || !round_coeffs_out || !final_evals_out)
			throw Error(Error::InputValidation, "null argument");
		if (n_vars == 0 || n_vars >= 40) throw Error(Error::InputValidation, "n_vars out of range");
		if (eq_ind_elems < ((uint64_t)1 << (n_vars - 1))) throw Error(Error::InputValidation, "the indicator's scratch holds fewer than 2^(n_vars - 1) elements");
		const bool prof = AbiProf::on(); // BNH_PROF=1: wall time of the set-up, of execute / fold over all rounds, of finish (diagnostic)
		auto now = [] { return std::chrono::steady_clock::now(); };
		const auto t_begin = now();
		ComputeLayer hal(ctx);
		Mi355xBackend backend(hal);
		DeviceBumpAllocator dev_alloc(FSliceMut{d_eq_ind, (size_t)eq_ind_elems});
		std::vector<SumcheckMultilinear> mls;
		for (uint32_t j = 0; j < n_mls; j++) mls.push_back(SumcheckMultilinear::folded(FSlice{d_multilins[j], (size_t)1 << n_vars}));
		std::vector<EqIndComposition> comps;
		std::vector<B128> sv;
		size_t at = 0, at_inf = 0;
		for (uint32_t c = 0; c < n_comps; c++) {
			EqIndComposition ec;
			ec.composition = hal.compile_expr(ArithCircuit::from_steps(std::vector<bn_step>(steps + at, steps + at + n_steps[c])));
			ec.composition_at_infinity = hal.compile_expr(ArithCircuit::from_steps(std::vector<bn_step>(steps_inf + at_inf, steps_inf + at_inf + n_steps_inf[c])));
			at += n_steps[c];
			at_inf += n_steps_inf[c];
			ec.degree = degrees ? degrees[c] : 2;
			comps.push_back(ec);
			sv.emplace_back(sums[c].lo, sums[c].hi);
		}
		std::vector<B128> eqc;
		for (uint32_t i = 0; i < n_vars; i++) eqc.emplace_back(eq_ind_challenges[i].lo, eq_ind_challenges[i].hi);
		// eq_ind_expand, High-to-Low: the tensor expansion of all challenges but the last (eq_ind.rs:430-446)
		const FSlice table = backend.tensor_product_full_query(std::vector<B128>(eqc.begin(), eqc.end() - 1), dev_alloc);
		EqIndSumcheckProver prover(hal, backend, dev_alloc, n_vars, std::move(mls), std::move(comps), std::move(sv), eqc, FSliceMut{const_cast<void *>(table.ptr), table.len_});
		const B128 bc(batch_coeff->lo, batch_coeff->hi);
		double us_exec = 0, us_fold = 0;
		const auto t_setup = now();
		for (uint32_t r = 0; r < n_vars; r++) {
			const auto t0 = prof ? now() : std::chrono::steady_clock::time_point{};
			const std::vector<B128> rc = prover.execute(bc);
			const auto t1 = prof ? now() : t0;
			for (size_t i = 0; i < rc.size(); i++) round_coeffs_out[rc.size() * r + i] = rc[i].raw();
			prover.fold(B128(challenges[r].lo, challenges[r].hi));
			if (prof) {
				us_exec += std::chrono::duration<double, std::micro>(t1 - t0).count();
				us_fold += std::chrono::duration<double, std::micro>(now() - t1).count();
			}
		}
		const auto t_rounds = now();
		const std::vector<B128> fin = prover.finish();
		for (size_t j = 0; j < fin.size(); j++) final_evals_out[j] = fin[j].raw();
		if (prof)
			fprintf(stderr, "[bnh prof] eq-ind sumcheck, %u rounds, %u multilinears, %u compositions: set-up %.1f us, execute %.1f us, fold %.1f us, finish %.1f us\n", n_vars,
			        n_mls, n_comps, std::chrono::duration<double, std::micro>(t_setup - t_begin).count(), us_exec, us_fold,
			        std::chrono::duration<double, std::micro>(now() - t_rounds).count());
		return 0;
	} catch (const Error &e) {
		g_err = e.what();
		return (int)e.kind();
	} catch (const std::exception &e) {
		g_err = e.what();
		return BN_ERR_CORE_LIB;
	}
}

} // extern "C"
