// binius_amd/csrc/internal.hpp -- declarations shared between the C-ABI layer (abi.cpp) and the
// kernel translation units.  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>
#include <memory>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/binius_amd.h"
#include "gf128.hpp"

struct bn_expr {
	std::vector<bn_step> steps;
	uint32_t n_vars = 0;
	// classification for the specialised kernels
	enum Shape { GENERIC = 0, PRODUCT = 1 } shape = GENERIC;
	std::vector<uint32_t> product_vars; // PRODUCT: var indices multiplied together (in order)
	mutable bn_step *d_steps = nullptr; // device copy for the interpreter kernels (uploaded on first use)
	int device = 0;
	// the circuit as a sum of monomials coeff * prod(vars) (abi_hal.cpp expand_poly), computed on first use by the old HAL's routed
	// paths: 0 = not yet, 1 = `poly` holds it, 2 = too large / not a polynomial the routed paths take
	struct monomial {
		std::vector<uint32_t> vars; // sorted; variables may repeat
		bn::f128 coeff;
	};
	mutable int poly_state = 0;
	mutable std::vector<monomial> poly;
};

namespace bn {
// ---- the widths the deferral machinery is sized for (one place; the static_asserts below tie them together)
constexpr int kFoldBatchMax = 32;    // arrays of ONE plain fold launch (fold_batch rides in the kernel-argument block) and of the single-claim slot bn_ctx::pend
constexpr int kFoldCallMax = 256;    // arrays of one bn_extrapolate_line_batch CALL (the shims' FOLD_BATCH_MAX): split into launches of kFoldBatchMax where it runs eagerly
constexpr int kGroupMaxArrays = 256; // multilinears of one prover on the claim-group path (abi_group.cpp) = arrays of one deferred group fold
constexpr int kGroupMaxClaims = 384; // product claims of one prover on that path (keccak: every committed column against up to three evaluation points)
constexpr int kGroupMaxJobs = 1024;  // jobs of one launch of kernels_group.hip (the table lives in pinned memory, not in the kernel arguments)
constexpr int kGroupMaxSlots = 1024; // accumulator slots of one launch: two per claim, the calling prover's and the riders'
constexpr int kGroupMaxGrid = 512;   // workgroups of one launch (the head-of-workgroup table in the kernel arguments)
static_assert(kFoldCallMax == kGroupMaxArrays, "a prover's fold arrives as ONE batch: hosted sessions fold whole batches");
static_assert(kFoldCallMax == BN_FOLD_CALL_MAX, "include/binius_amd.h states the limit the shims split at");
static_assert(kGroupMaxArrays <= kGroupMaxJobs && 2 * kGroupMaxClaims <= kGroupMaxSlots && kGroupMaxClaims <= kGroupMaxJobs, "group widths");
static_assert(kFoldBatchMax <= kFoldCallMax && kFoldBatchMax <= 32, "bn_ctx::pending_fold::scale_mask is a 32-bit mask over the batch");
constexpr int kPeerMaxWorld = 16; // ranks of one node that can share a peer exchange (finalize.hpp)
constexpr uint64_t kHtMaxM = 4096; // largest array (elements) a host tail can take over (abi_kernels.cpp)
// What a host tail (abi_kernels.cpp) leaves for the device to catch up with: the host folded its copy of two small arrays in place
// `levels` times; out[j][0 .. n0) -- where the FIRST of those folds wrote -- must end up holding what the host copy holds there.
struct tail_writeback_args {
	void *out[2];
	uint32_t n0;
};
}

struct bn_ctx {
	std::recursive_mutex mu; // serialises the entry points of this context
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	void *arena = nullptr;
	uint64_t arena_elems = 0;
	// device scratch owned by the context (partials of reductions, Local kernel buffers, ...)
	void *scratch = nullptr;
	size_t scratch_bytes = 0;
	bn::f128 *d_result = nullptr; // small result mailbox (256 elements)
	bn::f128 *h_result = nullptr; // pinned host mirror
	// zero-copy return path: kernels write returned scalars straight into fine-grained pinned host
	// memory and publish a sequence number; the host spins on it instead of memcpy + stream sync
	bn::f128 *h_mail = nullptr;        // host view   [0..64) values, slot 64 = sequence word
	bn::f128 *d_mail = nullptr;        // device view of the same memory
	uint64_t mail_seq = 0;
	bool s_clean = false;              // accumulator slots d_result[0..64) known to be zero
	uint8_t *d_mul8 = nullptr;         // 64 KiB GF(2^8) product table (tiled NTT)
	uint64_t *d_s_evals = nullptr;     // the twiddle basis last handed to bn_ntt_* / bn_fri_fold (BN_NTT_MAX_DIM^2 words) ...
	std::vector<uint64_t> h_s_evals;   // ... and its host copy: an NTT instance's basis is uploaded once, not per call
	unsigned *d_ticket = nullptr;      // device-scope ticket counter for the fused finalize
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	int n_cu = 256;
	// per-class kernel timing (bn_prof_begin / bn_prof_end)
	bool prof_on = false;
	struct prof_rec {
		int cls;
		hipEvent_t a, b;
	};
	std::vector<prof_rec> prof;
	std::vector<hipEvent_t> ev_pool;
	// deferred extrapolate_line batch (abi.cpp): launched by the next API call, or folded into the
	// next round-evaluation launch when that launch reads exactly the folded arrays
	struct pending_fold {
		bool active = false;
		uint32_t count = 0;
		uint64_t n = 0;
		bn::f128 z{0, 0};
		uint32_t scale_mask = 0; // bit i: the upper half of folded array i is multiplied by hi_scale (bn_extrapolate_line_batch_scaled)
		bn::f128 hi_scale{0, 0};
		void *x0[bn::kFoldBatchMax] = {};        // evals_0: written in place ...
		const void *x1[bn::kFoldBatchMax] = {};
		const void *src0[bn::kFoldBatchMax] = {}; // ... and read from here (== x0 unless a deferred copy_d2d fed it)
	} pend;
	// a SECOND deferred fold, chained in place on pend's output: exists only between the call that answered a round from
	// the precomputed sums below and the next round evaluation, which then folds twice (kernels_foldeval8.hip)
	pending_fold pend2;
	// the next round's evaluations as quadratics in the next challenge, left by a two-round launch
	struct precomp_state {
		bool valid = false, consumed = false;
		const void *lo[2] = {}, *hi[2] = {}; // the halves of the two arrays the sums describe ...
		uint64_t m = 0;                      // ... m elements each
		bn::f128 P0{0, 0}, P1v{0, 0}, P2{0, 0}, Q0{0, 0}, Q1v{0, 0}, Q2{0, 0}; // coefficient applied; P1v = P(1), Q1v = Q(1)
		std::vector<unsigned char> recipe;   // of the caller's two-sum request
	} pre;
	// Weighted shadow of the MLE-check round evaluation (abi_kernels.cpp "shadow"): the literal call sequence of
	// BivariateMLEcheckProver asks for sum a * b * eq every round; the backend keeps S = lambda * b (.) eq beside the caller's
	// arrays and answers from the bivariate kernels on (a, S).
	struct shadow_state {
		bool valid = false;
		const void *a_lo = nullptr, *a_hi = nullptr; // the halves the NEXT evaluation will pass for the plain factor ...
		const void *b_lo = nullptr, *b_hi = nullptr; // ... and for the factor whose weighted copy S is kept
		uint64_t half = 0;                           // elements per half (= rows of the next evaluation)
		const void *eq = nullptr;                    // the table the next evaluation will pass, eq_len entries
		uint64_t eq_len = 0;
		const void *eq_copy_src = nullptr;           // a copy_d2d of the table's lower half seen since the last fold: src -> dst
		void *eq_copy_dst = nullptr;
		void *S = nullptr;                           // 2 * half elements (lower | upper half), own allocation
		uint64_t S_cap = 0;                          // elements allocated
		bn::f128 lambda{1, 0};                       // S = lambda * b (.) eq on both halves
		std::vector<bn::f128> rho_inv, one_minus_zeta; // per variable k of the ORIGINAL table (bit k of its index)
		bool checked = false;                        // the table's tensor structure has been verified and the ratios derived
		uint64_t blocked_below = 0;                  // a table failed the check at this many rows: no new shadow for the smaller rounds of that instance
		bool fold_pending = false;                   // the caller's fold of (a, b) is deferred in pend; S folds with it
		bn::f128 z{0, 0}, hi_scale{0, 0}, lambda_next{1, 0};
		uint32_t ia = 0, ib = 1;                     // which array of the pending batch is a / b
	} shadow;
	// side stream: work of a shadowed MLE-check that nothing on the main stream depends on (the fold of b, the folds of the
	// indicator table) runs beside the sumcheck's kernels; every flush joins it back (abi.cpp side_stream / side_join)
	hipStream_t side = nullptr;
	hipEvent_t side_ev = nullptr, main_ev = nullptr;
	bool side_busy = false;
	// side work that has been asked for but not launched yet: a launch costs the host ~3 us, so it is issued while the host
	// would otherwise spin on the round's result (right after the round's kernel has been launched or signalled), in order
	struct side_op {
		enum { COPY, ADD_ASSIGN, ADD, FOLD } kind; // FOLD: dst = evals_0 (written), src = evals_1, src2 = where evals_0 is read (nullptr: dst)
		void *dst;
		const void *src, *src2;
		uint64_t n;
		bn::f128 z;
	};
	std::vector<side_op> side_queue;
	unsigned *d_flag = nullptr;  // one word of device memory for yes/no answers of checking kernels
	bool shadow_enabled = true;  // BN_MLECHECK_SHADOW=0 turns the shadow off
	uint64_t shadow_created = 0, shadow_rounds = 0, shadow_dropped = 0;
	bool two_round = true;       // BN_TWO_ROUND=0 turns the two-round launches off
	uint64_t two_round_hosted = 0, two_round_launches = 0; // rounds answered by the host from the sums / launches of the kernel
	// deferred copy_d2d (the "allocate a new buffer for the folded evaluations and copy in evals_0"
	// of the first fold, v3/bivariate_product.rs:196-206): absorbed by the fold that overwrites its
	// destination, which then reads evals_0 from the copy's source
	struct pending_copy {
		const void *src;
		void *dst;
		uint64_t n;
	};
	std::vector<pending_copy> pend_copies;
	// host mirror of tiny folded arrays (k_fold_publish): valid until the next call that may write
	// device memory; mailbox slot = off + i for element i of [ptr, ptr + n)
	struct mirror_state {
		bool valid = false;
		bool host = false; // the values were computed on the host (host_vals): nothing to wait for
		uint64_t seq = 0;
		uint32_t count = 0, n = 0;
		const void *ptr[bn::kFoldBatchMax] = {};
		bn::f128 host_vals[bn::kFoldBatchMax] = {};
	} mirror;
	// The four elements per array that the LAST two-round launch of a sumcheck leaves (kernels_foldeval8.hip publishes them
	// beside its sums): the two folds that remain are six host products, so the caller's read of the final evaluations does
	// not wait for the kernel that performs them on the device (it is launched all the same: memory ends up as always).
	struct final_y_state {
		bool valid = false;
		const void *lo[2] = {};
		bn::f128 y[2][4] = {};
	} fin_y;
	// Host tail of a sumcheck (abi_kernels.cpp "host tail"): once the arrays are down to a few hundred elements the
	// two-round kernel hands them to the host (already mapped into the power basis of hostmul_clmul.cpp) and the remaining
	// rounds -- evaluations and folds -- are host arithmetic: no launch, no round trip.  The device catches up with ONE
	// launch that writes the host's folded copy back (launch_tail_writeback) when the caller reads the final evaluations or does anything
	// else; the caller's memory ends up exactly as eager execution leaves it.
	struct host_tail_state {
		bool active = false;
		bool evaluated = false;        // the evaluation of the current arrays has been answered: the next expected call is their fold
		uint32_t n_levels = 0;         // folds performed on the host and not yet on the device
		bn::tail_writeback_args chain{}; // ... and where their results belong
		uint64_t cur_m = 0;            // elements per array of the current (host) arrays
		const void *cur_lo[2] = {}, *cur_hi[2] = {}; // device addresses of their halves: what the next calls must name
		std::vector<uint64_t> y[2];    // the arrays in the power basis, 2 words per element
	} ht;
	void *h_tail = nullptr, *d_tail = nullptr; // pinned staging (host / device view): [0, 2 * kHtMaxM) the kernel mirrors Y into, then kHtMaxM elements of the host's folded copy for the write-back
	uint64_t *d_ht_tag = nullptr;              // the tag accumulator of host-tail launches of several workgroups (zero between launches)
	void *d_phi = nullptr;                     // nibble tables of the basis change and of its inverse (2 x 8 KiB of device memory)
	bool ht_enabled = false;                   // BN_HOST_TAIL=0 turns it off; needs PCLMULQDQ on the host
	bool ht_peer_ok = false;                   // bn_host_tail_allow_peer: the caller exchanges the host rounds' partials itself
	uint64_t ht_max = 256;                     // largest Y (elements per array) the host takes over (BN_HOST_TAIL_MAX_LOG2 <= 12; default 2^12 when the host folds on VPCLMULQDQ, else 2^8)
	uint64_t ht_started = 0, ht_rounds = 0, ht_flushed = 0; // instances taken over, evaluations answered, chains launched
	bool circuit_multipass = true; // BN_CIRCUIT_MULTIPASS=0: generic circuits stay on the scalar interpreter kernels (abi_circuit.cpp)
	void *ntt_cache = nullptr; // bn::ntt_bs_cache (allocated on first use)
	// all-ones | all-zeros tables of the old HAL's routed round evaluation (abi_hal.cpp): filled once per size, kept
	void *hal_const = nullptr;
	uint64_t hal_const_half = 0; // elements per table
	// wide constraint-set requests of the old HAL (abi_hal.cpp round_evals_eq_set): the element-wise products' job table (pinned,
	// device-mapped) and the plan of the last request (which monomials, which columns are scaled by the indicator), reused while
	// the same compiled compositions come back round after round
	bool hal_eq_set = true; // BN_HAL_EQ_SET=0: such requests are dealt out to parts of the general code instead
	bool hal_coef = true;   // BN_HAL_COEF=0: degree-3 / domain-point requests keep the general code (rows + compiled circuits)
	void *h_mul_jobs = nullptr, *d_mul_jobs = nullptr;
	std::shared_ptr<void> hal_set_plan;
	// pinned, device-mapped staging of bn_gather_d2h: offsets in, gathered items out (grown on demand)
	void *h_gather = nullptr, *d_gather = nullptr;
	size_t gather_bytes = 0;
	bool lazy_fold = true; // BN_NO_LAZY_FOLD=1 turns the deferral off
	// resident tail kernel (kernels_foldeval9.hip k_foldeval_tail, protocol in abi.cpp)
	struct tail_state {
		bool active = false;
		uint64_t id = 0;          // launch counter; commands are (id << 20) | round
		uint64_t round = 0;       // rounds already executed after the first
		uint64_t n_in_next = 0;   // size of the next round's (pre-fold) arrays
		void *out[2] = {nullptr, nullptr};
		uint64_t seq0 = 0;        // mailbox sequence of the first round
		std::vector<unsigned char> recipe; // bytes of the fin_args the kernel was launched with (seq zeroed)
	} tail;
	uint64_t tail_counter = 0;
	uint64_t tail_max_n_in = 0; // BN_TAIL_MAX_LOG2=3..12 enables the resident tail (off by default: see DESIGN.md)
	// armed round (arm.hpp): the kernel of the NEXT small round, enqueued behind the current one and waiting for its
	// challenge on the command block h_mail[84..87]
	struct arm_state {
		bool active = false;
		uint64_t id = 0;
		uint64_t n_in = 0;
		const void *x0[2] = {}, *x1[2] = {};
		void *out[2] = {};
		uint32_t scale_mask = 0;
		uint64_t seq = 0;      // the mailbox sequence number it will publish
		uint64_t peer_round = 0; // the peer-exchange round it will take part in (0: none)
		uint32_t nf = 0;         // 0: a one-round kernel (k_foldeval9*, k_foldeval_mfma); 1, 2: k_foldeval8 with that many folds
		bn::f128 *d_sums = nullptr; // its accumulator slots
		std::vector<unsigned char> recipe;
	} arm;
	uint64_t arm_counter = 0;
	uint64_t *d_arm_relay = nullptr; // device memory, 8 words
	bool arm_enabled = true;         // BN_ARM=0 turns the armed rounds off
	uint64_t arm_hits = 0, arm_cancels = 0, arm_expired = 0;
	uint64_t arm_ns_parse = 0; // entry of bn_kernel_launch -> challenge handed over
	uint64_t arm_ns_wait = 0, arm_ns_launch = 0; // go -> mailbox seen; of which: enqueueing the next armed kernel
	// ---- claim groups (abi_group.cpp, kernels_group.hip): any number of deferred folds, each a prover's batch, beside the legacy
	// single slot `pend`; the product claims of every prover that is ready evaluated in ONE launch, the others' sums kept for
	// their execute().  The unit is the PROVER (a session: its arrays and claims), not the context.
	struct group_fold {
		uint32_t count = 0;
		uint64_t n = 0; // elements per half = elements of the folded array
		bn::f128 z{0, 0};
		std::vector<void *> x0;         // count entries each (at most bn::kGroupMaxArrays)
		std::vector<const void *> x1, src0;
		const char *hull_b = nullptr, *hull_e = nullptr; // every byte the batch reads or writes lies in [hull_b, hull_e): the quick "does not touch" test
		void push(void *o, const void *hi, const void *lo) // (n is set)
		{
			x0.push_back(o);
			x1.push_back(hi);
			src0.push_back(lo);
			count++;
			for (const void *p : {(const void *)o, hi, lo}) {
				const char *b = (const char *)p, *e = b + n * sizeof(bn::f128);
				if (!hull_b || b < hull_b) hull_b = b;
				if (!hull_e || e > hull_e) hull_e = e;
			}
		}
		bool hull_hits(const void *p, uint64_t elems) const { return (const char *)p < hull_e && hull_b < (const char *)p + elems * sizeof(bn::f128); }
	};
	struct group_session {
		uint32_t m = 0, k = 0;                      // arrays (<= bn::kGroupMaxArrays), product claims (<= bn::kGroupMaxClaims)
		uint64_t row_len = 0;                       // points of the last evaluation
		std::vector<const void *> lo, hi;           // [m] the arrays' halves at the last evaluation (what the prover's next fold reads)
		std::vector<uint16_t> pa, pb;               // [k] the claims, as indices into the arrays
		bool pre_valid = false;                     // the raw sums of the NEXT evaluation were computed ahead ...
		uint64_t pre_row_len = 0;                   // ... of these halves
		std::vector<const void *> pre_lo, pre_hi;   // [m]
		const char *pre_hull_b = nullptr, *pre_hull_e = nullptr; // hull of those halves
		std::vector<bn::f128> pre_raw;              // [2 k] claim c: [2 c] at 1, [2 c + 1] at infinity
		uint64_t stamp = 0;
		// hosted: the prover's arrays are small enough that its remaining rounds are host arithmetic (abi_group.cpp "hosted
		// sessions"): hy[j] = array j in the power basis of hostmul_clmul.cpp (2 words per element), folded in place exactly as
		// the device would; the device catches up with ONE launch (write-back of the first h_n0 elements per array to h_out[j],
		// where the first host fold wrote) when anybody looks at the memory
		bool hosted = false;
		uint64_t h_len = 0;                         // elements per array now
		std::vector<const void *> h_lo, h_hi;       // [m] device addresses of the current arrays' halves: what the next calls must name
		uint32_t h_levels = 0;                      // folds performed on the host and not yet on the device
		uint64_t h_n0 = 0;
		std::vector<void *> h_out;                  // [m]
		std::vector<std::vector<uint64_t>> hy;
	};
	struct group_state {
		bool enabled = true;   // BN_GROUP=0: every call takes the single-claim machinery / the eager kernels
		bool speculate = true; // BN_GROUP_SPEC=0: a launch only carries the calling prover's claims
		uint64_t chain_min_rows = (uint64_t)1 << 21; // evaluation points per claim from which the jobs of a prover that depend on each other are chained
		                                              // inside the launch; below: shared arrays are folded by a plain launch in front (BN_GROUP_CHAIN_MIN_LOG2; 63: never)
		bool on = false;       // a group fold or evaluation happened since the last full flush: single-claim requests join in
		std::vector<group_fold> folds;
		std::vector<group_session> sessions;
		uint64_t stamp = 0;
		uint64_t launches = 0, jobs_fused = 0, jobs_eval = 0, prefolds = 0, spec_jobs = 0, spec_hits = 0, evals = 0, flushed_folds = 0, jobs_fold = 0, chain_count = 0;
		uint64_t ht_max = 0;   // largest array (elements) a hosted session starts with (0: off; BN_GROUP_HT_MAX_LOG2)
		uint64_t ht_work = 8 * 4096; // ... and the most elements over all its arrays (bn::kGroupTailWorkElems; BN_GROUP_HT_WORK_LOG2)
		void *h_stage = nullptr, *d_stage = nullptr; // pinned staging, 2 x kGroupTailMaxElems elements: hand-over | write-back
		uint64_t hosted_started = 0, hosted_evals = 0, hosted_folds = 0, hosted_writebacks = 0;
		// what a launch of kernels_group.hip reads besides its kernel arguments -- the job table, the pointer tables of a hand-over and
		// of a write-back -- lives in ONE pinned, device-mapped block (bn::group_tables): written by the host right before the launch,
		// read by the workgroups over the bus.  Every launch that reads the job table or the hand-over's pointers is awaited by its
		// caller (the mailbox) before the next one is prepared; the write-back's pointers are rewritten only behind a stream
		// synchronisation (abi_group.cpp unhost).
		void *h_tables = nullptr, *d_tables = nullptr;
		// BN_GROUP_PROF=1 (diagnostic): host nanoseconds by phase of the group path, printed when the context is destroyed
		enum { P_PARSE = 0, P_MATCH, P_PLAN, P_LAUNCH, P_WAIT, P_ANSWER, P_HOSTED, P_DEFER, P_HOST_WAIT, P_HOST_COPY, P_HOST_FOLD, P_N };
		bool prof = false;
		uint64_t prof_ns[P_N] = {}, prof_calls[P_N] = {};
		bn::f128 *d_S = nullptr;                     // kGroupMaxSlots accumulator slots (zero between launches)
		bn::f128 *h_gmail = nullptr, *d_gmail = nullptr; // pinned: the raw sums of a launch, kGroupMaxSlots values (the sequence word stays in h_mail[64])
	} grp;
	// cross-rank reduction inside the finalize step (bn_peer_*, finalize.hpp peer_exchange)
	struct peer_state {
		uint32_t world = 0, rank = 0;
		void *own = nullptr;                       // this rank's mailbox (fine-grained device memory)
		void *box[bn::kPeerMaxWorld] = {};         // every rank's mailbox as mapped here (box[rank] == own)
		bool connected = false, active = false;    // active: reduce the round evaluations launched from now on
		uint32_t stress = 0;                       // BN_PEER_STRESS, read when the mailbox is created
		uint64_t round = 0;                        // rounds executed so far (monotonic for the life of the attachment)
	} peer;
};

namespace bn {

constexpr int kResultSlots = 256;
// bumped by every bn_expr_free: whatever remembers compiled expressions by address (abi_hal.cpp's plans) forgets them
uint64_t expr_epoch();
void expr_epoch_bump();

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);
int hip_fail(hipError_t e, const char *what);
void *ctx_scratch(bn_ctx *ctx, size_t bytes); // grows on demand; nullptr on failure

#define BN_HIP(expr)                                   \
	do {                                               \
		hipError_t _e = (expr);                        \
		if (_e != hipSuccess)                          \
			return bn::hip_fail(_e, #expr);            \
	} while (0)

// ---- kernels_stream.hip
hipError_t launch_fill(hipStream_t s, void *dst, uint64_t n, f128 v);
hipError_t launch_add_assign(hipStream_t s, void *dst, const void *src, uint64_t n);
hipError_t launch_add(hipStream_t s, void *dst, const void *src1, const void *src2, uint64_t n);
hipError_t launch_extrapolate_line(hipStream_t s, int n_cu, void *evals_0, const void *evals_1, uint64_t n, f128 z);
struct fold_batch {
	void *x0[kFoldBatchMax];
	const void *x1[kFoldBatchMax];
	const void *src0[kFoldBatchMax]; // nullptr: in place (evals_0 is read from x0); else evals_0 is read from here and written to x0
};
hipError_t launch_extrapolate_line_batch(hipStream_t s, int n_cu, const fold_batch &b, uint32_t count, uint64_t n, f128 z);
constexpr int kFoldWideMax = 128; // arrays of one wide fold launch (the batch rides in the kernel arguments: 3 KiB of the 4)
struct fold_batch_wide {
	void *x0[kFoldWideMax];
	const void *x1[kFoldWideMax];
	const void *src0[kFoldWideMax]; // as fold_batch
};
hipError_t launch_extrapolate_line_wide(hipStream_t s, int n_cu, const fold_batch_wide &b, uint32_t count, uint64_t n, f128 z);
struct fold_lengths {
	uint64_t n[kFoldBatchMax];
};
// arrays of different lengths under one challenge (the folds of several provers of a batch round in one launch)
hipError_t launch_extrapolate_line_ragged(hipStream_t s, int n_cu, const fold_batch &b, const fold_lengths &fl, uint32_t count, f128 z);
hipError_t launch_fold_publish(hipStream_t s, void *const *x0, const void *const *src0, const void *const *x1, uint32_t count, uint32_t n,
                               f128 z, f128 *d_mail, uint64_t seq, uint32_t scale_mask = 0, f128 hi_scale = f128{0, 0});
hipError_t launch_scale(hipStream_t s, int n_cu, void *x, uint64_t n, f128 c); // x[i] *= c
hipError_t launch_xor_sum(hipStream_t s, int n_cu, const void *x, uint64_t n, f128 *d_out); // d_out[0] ^= XOR_i x[i]
hipError_t launch_scale_to(hipStream_t s, int n_cu, void *out, const void *x, uint64_t n, f128 c); // out[i] = c * x[i]
hipError_t launch_tensor_expand(hipStream_t s, int n_cu, void *data, uint32_t log_n, const f128 *coords, uint32_t k);

// ---- kernels_roundeval.hip
struct fin_term {
	uint32_t value;
	uint32_t slot;
	f128 coeff;
};
// fused round evaluation of a product composition: for factor j, hi[j] are the evaluations at 1;
// lo[j] != null means the evaluation at infinity is lo[j]+hi[j], lo[j] == null means the factor is
// the same at both points (e.g. the eq-indicator).  d_out[0] ^= S_1, d_out[1] ^= S_inf (unscaled).
struct fin_fuse; // defined below: finalize work for the last workgroup of the round-eval kernel
hipError_t launch_roundeval_product(hipStream_t s, int n_cu, const void *const *hi, const void *const *lo, uint32_t k,
                                    uint64_t n, f128 *d_out, const fin_fuse *fuse);
constexpr int kFinMaxTerms = 32, kFinMaxValues = 8, kFinMaxRets = 8;
// passed BY VALUE as a kernel argument: no host->device staging copy on the per-round path
struct fin_args {
	uint32_t n_terms, n_values, n_ret, n_slots; // n_slots accumulator slots are re-zeroed after use
	uint64_t seq;                                // != 0: publish rets + seq to the host mailbox
	fin_term terms[kFinMaxTerms];
	f128 init[kFinMaxValues];
	uint32_t ret_ids[kFinMaxRets];
};
// values[v] = init[v] ^ XOR_t coeff_t * S[slot_t], then rets[i] = values[ret_ids[i]]
struct fin_peer; // below: cross-rank reduction of the returned values (nullptr / world <= 1: none)
hipError_t launch_finalize(hipStream_t s, const fin_args &args, f128 *d_S, f128 *d_rets, f128 *d_mail, const fin_peer *peer = nullptr);
hipError_t launch_xor_publish(hipStream_t s, const f128 *d_vals, uint32_t n_groups, uint32_t group_len, f128 *d_rets, f128 *d_mail,
                              uint64_t seq, uint32_t g_stride = 0, uint32_t i_stride = 0); // (strides 0: g_stride = group_len, i_stride = 1)
// Cross-rank reduction of the returned values inside the finalize step (sharded sumcheck, SURVEY.md section 8e;
// bn_peer_*): every rank owns a mailbox in fine-grained device memory that all ranks of the node have mapped
// (hipIpc; over xGMI between devices).  The finalizing workgroup stores this rank's returned values into its slot of
// EVERY rank's mailbox, waits until all `world` slots of its own mailbox carry this round's number, and publishes
// the XOR.  Mailbox = [2 (round parity)][kPeerMaxWorld (writer rank)][kPeerSlotWords] 64-bit words; slot words
// 0 .. 2 n_ret - 1 = values, word 16 = the slot's tag = a 64-bit mix of the round number and the value words: a reader accepts
// a slot only when the tag it read is the tag of the values it read (finalize.hpp peer_exchange).
constexpr int kPeerSlotWords = 24;
constexpr size_t kPeerMailboxBytes = 2 * kPeerMaxWorld * kPeerSlotWords * sizeof(uint64_t);
struct fin_peer {
	uint64_t *box[kPeerMaxWorld]; // device-visible mailbox of every rank (box[rank] = this rank's own)
	uint32_t world, rank;         // world <= 1: no exchange
	uint64_t round;               // >= 1, the same on every rank, +1 per reduced launch
	uint32_t stress;              // BN_PEER_STRESS (test only): bit 0 = tag before the values, bit 1 = pauses between the value words
};
// fused form: the last workgroup to finish (device-scope ticket counter) runs the finalize body
struct fin_fuse {
	fin_args args;
	f128 *S;
	f128 *rets;
	f128 *mail;
	unsigned *counter; // zero before the launch; the last workgroup resets it
	fin_peer peer;
};
// generic: sum_i C(rows[0][i], ..) over a circuit, XOR-accumulated into d_out[0]
hipError_t launch_sum_composition_generic(hipStream_t s, int n_cu, const void *const *d_rows_dev, uint32_t n_rows,
                                          uint64_t row_len, const bn_step *d_steps, uint32_t n_steps, f128 *d_out);
// elementwise product sum over k rows (k>=1), bit-sliced: d_out[0] ^= sum_i prod_j rows[j][i]
hipError_t launch_sum_product(hipStream_t s, int n_cu, const void *const *rows, uint32_t n_rows, uint64_t row_len,
                              f128 *d_out);

// ---- kernels_roundeval9.hip (the hot bivariate-product kernel)
hipError_t launch_roundeval9_pair(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi,
                                  const void *b_lo, uint64_t n, f128 *d_out, const fin_fuse *fuse);
// fold + next round evaluation in one pass (kernels_foldeval9.hip): per array, where the two halves
// are read and where the folded half is written
struct foldeval_args {
	const void *x0[2];
	const void *x1[2];
	void *out[2];
	// bn_extrapolate_line_batch_scaled: bit j set -> the upper half of out[j] (the elements the next round reads as
	// "evaluation at 1") is multiplied by hi_scale after the fold
	uint32_t scale_mask;
	uint32_t xcd_tiles; // k_foldeval_mfma: XCD-contiguous tile order (set by the launcher)
	f128 hi_scale;
};
bool foldeval9_is_small(int n_cu, uint64_t n_in);
struct arm_args; // arm.hpp: non-null = an armed launch (z and hi_scale arrive through the command block)
// kernels_foldeval_fp4.hip: the fused fold + evaluation with fold waves and FP4 Gram waves (plain fold, whole tiles)
struct fin_fuse;
bool foldeval_fp4_applies(int n_cu, const foldeval_args &fa, uint64_t n_in);
hipError_t launch_foldeval_fp4(hipStream_t s, int n_cu, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse &fz,
                               const arm_args &arm, bool nt);
hipError_t launch_foldeval9(hipStream_t s, int n_cu, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse *fuse,
                            const arm_args *armed = nullptr);
hipError_t launch_foldeval_tail(hipStream_t s, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse &fz,
                                const uint64_t *d_cmd, uint64_t *d_status, uint64_t tail_id);
// ---- kernels_foldeval8.hip: two rounds per launch for the small rounds -- 0, 1 or 2 folds, then the eight quarter sums
// that answer this round AND (as a quadratic in the next challenge) the next one
struct foldeval8_args {
	const void *x0[2]; // n_folds >= 1: lower / upper half of the arrays before the first fold (n_in elements);
	const void *x1[2]; // n_folds == 0: lower / upper half of Y itself
	void *out[2];      // n_folds >= 1: where Y (n_in >> n_folds elements) is written (may be x0)
	uint64_t n_in;
	uint32_t n_folds;
	// host tail (abi_kernels.cpp): non-null -> Y is also handed to the host, mapped into the host's power basis on the way
	// (mirror[arr * M + i] = Phi(Y_arr[i]), pinned host memory; phi_tab = the 512-entry nibble table of Phi in device
	// memory, ctable.hpp layout).  tag_acc: a zeroed 64-bit word of device memory in which the workgroups of a launch of several
	// accumulate the staging's tag (the last one publishes and re-zeroes it); may be null for a single workgroup.
	f128 *mirror;
	const uint4 *phi_tab;
	uint64_t *tag_acc;
};
hipError_t launch_foldeval8(hipStream_t s, const foldeval8_args &fa, f128 z1, f128 z2, f128 *d_out8, const fin_fuse *fuse,
                            const arm_args *armed = nullptr);
// the last two folds of a sumcheck in one launch (4 n_out -> n_out elements per array, count * n_out <= 64), mirrored into
// the mailbox like launch_fold_publish
// the device catching up with a host tail: out[j][i] = PhiInv(staging[j * n0 + i]), i < n0 <= 2048 -- the host's folded copy (power
// basis, pinned host memory) mapped back to the tower basis through the nibble table of the inverse basis change
hipError_t launch_tail_writeback(hipStream_t s, const tail_writeback_args &a, const void *d_staging, const void *d_phi_inv);
hipError_t launch_fold2_publish(hipStream_t s, void *const *out, const void *const *src0, const void *const *x1, uint32_t count, uint32_t n_out,
                                f128 z1, f128 z2, f128 *d_mail, uint64_t seq);
hipError_t launch_roundeval9_eq(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo,
                                const void *eq, uint64_t n, f128 *d_out, const fin_fuse *fuse);
hipError_t launch_roundeval9_split(hipStream_t s, int n_cu, const void *a, const void *b, uint64_t n, uint64_t split_off,
                                   f128 *d_out);

// ---- kernels_roundeval_mfma.hip / kernels_foldeval_mfma.hip: the same three jobs with the products on the
// matrix cores (gram.hpp).  mfma_applies(): enough points to fill the chip with 256-point tiles (the
// 9-lane VALU kernels keep the small, latency-shaped rounds); BN_EVAL=valu forces the VALU kernels.
bool mfma_applies(int n_cu, uint64_t n_points);
hipError_t launch_roundeval_mfma_pair(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo,
                                      uint64_t n, f128 *d_out, const fin_fuse *fuse);
// the same sums with operands whose pairs are INTERLEAVED (Low-to-High evaluation order of the old HAL: e0 = x[2 i],
// e1 = x[2 i + 1]): log2 of the element stride per operand (0: unit stride as above, 1: every other element)
hipError_t launch_roundeval_mfma_pair_strided(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, uint32_t a_shift, const void *b_hi,
                                              const void *b_lo, uint32_t b_shift, uint64_t n, f128 *d_out);
// the same sums on the FP4 matrix path (kernels_roundeval_fp4.hip): round 0 of a sumcheck
hipError_t launch_roundeval_fp4_pair(hipStream_t s, int n_cu, const void *a_hi, const void *a_lo, const void *b_hi, const void *b_lo,
                                     uint64_t n, f128 *d_out, const fin_fuse *fuse);
hipError_t launch_roundeval_fp4_split(hipStream_t s, int n_cu, const void *a, const void *b, uint64_t n, uint64_t split_off, f128 *d_out);
hipError_t launch_roundeval_mfma_split(hipStream_t s, int n_cu, const void *a, const void *b, uint64_t n, uint64_t split_off, f128 *d_out);
hipError_t launch_foldeval_mfma(hipStream_t s, int n_cu, const foldeval_args &fa, uint64_t n_in, f128 z, f128 *d_out, const fin_fuse *fuse,
                                const arm_args *armed = nullptr);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, DEVICE): raised once per pair and process (and again if a
// later launch asks for more) -- a second context on another device of the same process gets its own (kernels_stream.hip)
hipError_t func_lds_limit(const void *fn, int bytes);

// ---- kernels_group.hip: the product claims of a whole batch round of sumchecks as jobs of ONE launch; returns raw sums
struct group_job {
	const void *x0[2], *x1[2]; // kind 0: lower / upper half of the two arrays BEFORE the fold (2 n elements each); kind 1: lower half (evaluations
	                           // at 0) / upper half (evaluations at 1) of the two arrays as they are (n elements each)
	void *out[2];              // kind 0: where the folded arrays (2 n elements each) are written (may be x0)
	f128 z;                    // kind 0: the fold's challenge
	uint64_t n;                // evaluation points of the job
	uint32_t kind;             // 0 = fold + evaluate, 1 = evaluate, 2 = two plain inner products: rows x0[0] . x0[1] -> S[slot], x1[0] . x1[1] -> S[slot + 1] (x1 null: one),
	                           // 3 = fold only: arrays x0[sd] / x1[sd] -> out[sd] as kind 0, no sums (x0[1] null: one array)
	uint32_t slot;             // the job's sums are XORed into S[slot] (at 1) and S[slot + 1] (at infinity)
	uint32_t wg_begin, wg_count; // (filled in by the launcher; wg_count = 0: a follower of the chain in front of it)
	uint32_t chain;            // the `chain` jobs that follow this one in the table are run by THIS job's workgroups, one after the other, each on
	                           // the tiles the workgroup had in this job (all jobs of a chain: the same n)
	uint32_t acquire;          // 1: the job reads what earlier jobs of its chain wrote (a workgroup reads back only its own tiles: its stores are
	                           // awaited and its vector cache invalidated first)
};
// hosted provers (abi_group.cpp): hand-over of a prover's arrays to the host (with its deferred fold performed on the way) and the
// write-back of the host's folded copies
constexpr uint64_t kGroupTailMaxElems = 32 * 4096; // elements of one hand-over / write-back: the pinned staging holds that many
// ... and a prover is handed over once its arrays together are at most this many elements: the host's rounds cost products in
// proportion (claims x points), the device's a launch each -- eight arrays of 4096 elements or a hundred of 256 (measured, 50
// claims over 100 arrays at 2^22: 10.1 ms from 256 elements per array on, 10.9 ms from 1024)
constexpr uint64_t kGroupTailWorkElems = 8 * 4096;
// the pinned block behind bn_ctx::group_state::h_tables / d_tables
struct group_ptr_table {
	const void *src0[kGroupMaxArrays]; // x1[j] != null: the fold's inputs (n elements each), its output goes to out[j]; x1[j] == null: the array itself
	const void *x1[kGroupMaxArrays];
	void *out[kGroupMaxArrays];
};
struct group_tables {
	group_job jobs[kGroupMaxJobs]; // of the launch in flight, sorted by first workgroup
	group_ptr_table mirror;        // of the hand-over in flight
	group_ptr_table writeback;     // of the last write-back (only `out` is used)
};
struct group_mirror_args {
	const group_ptr_table *ptrs; // device view of the pinned pointer table
	f128 z;
	uint32_t count, n;   // arrays, elements per array handed over
	f128 *staging;       // pinned host memory (device view): staging[j * n + i] = Phi(y_j[i])
	const uint4 *phi_tab; // nibble table of Phi (device memory, ctable.hpp layout)
	uint64_t *tag_acc;   // zeroed 64-bit word of device memory
	unsigned *counter;   // zeroed ticket
	f128 *mail;          // pinned mailbox: word 66 = tag, word 64 = seq
	uint64_t seq;
};
hipError_t launch_group_mirror(hipStream_t s, const group_mirror_args &a);
struct group_writeback_args {
	const group_ptr_table *ptrs; // device view; out[j] = where array j's first n0 elements belong
	uint32_t count, n0;
	const f128 *staging;  // pinned host memory (device view): the host's copies, power basis
	const uint4 *phi_inv; // nibble table of the inverse basis change
};
hipError_t launch_group_writeback(hipStream_t s, const group_writeback_args &a);
// jobs_in[0 .. n_jobs) (n_jobs <= kGroupMaxJobs): dealt out to workgroups, sorted and written to h_table (pinned; d_table = its device view),
// which the launch reads.  The raw sums S[0 .. n_slots) (n_slots <= kGroupMaxSlots; zero before, zero after) go to d_vals[0 .. n_slots),
// then the sequence number to d_mail[64].
hipError_t launch_group(hipStream_t s, int n_cu, const group_job *jobs_in, uint32_t n_jobs, uint32_t n_slots, f128 *d_S, f128 *d_vals, f128 *d_mail,
                        unsigned *d_counter, uint64_t seq, group_job *h_table, const group_job *d_table);

// ---- kernels_hal.hip: general forms of the old HAL's round calculation and lerp fold (abi_hal.cpp)
constexpr int kHalMaxMl = 16, kHalMaxEv = 8, kHalMaxPts = 13;
struct hal_round_args {
	struct ml_t {
		const uint4 *evals;
		uint64_t len;
		f128 suffix;
	} ml[kHalMaxMl];
	struct ev_t {
		const bn_step *steps, *steps_inf;
		uint32_t n_steps, n_steps_inf;
		uint32_t pt_start, pt_end;
		const uint4 *eq;
		uint32_t out_off;
	} ev[kHalMaxEv];
	f128 pts[kHalMaxPts]; // evaluation points of index 3, 4, ...
	uint32_t n_ml, n_ev, pt_lo, pt_hi, order, n_vars;
};
hipError_t launch_hal_round_evals(hipStream_t s, int n_cu, const hal_round_args &a, f128 *d_out);
hipError_t launch_hal_fold_lerp(hipStream_t s, int n_cu, const void *evals, uint64_t len, f128 suffix, uint32_t order, uint64_t half, f128 z,
                                void *out, uint64_t n_out);
// rows of the general round calculation (abi_hal.cpp): out[i] = the multilinear's value at evaluation point `point` of pair i
// -- 0: e0, 1: e1, 2 (infinity): e0 + e1, otherwise e0 + z (e0 + e1) --, pairs taken as in launch_hal_fold_lerp
// the same for up to kHalMaxRows (multilinear, point) pairs in one launch
constexpr int kHalMaxRows = 24;
struct hal_rows_args {
	struct job {
		const uint4 *evals;
		uint4 *out;
		uint64_t len;
		f128 suffix, z;
		uint32_t point;
	} jobs[kHalMaxRows];
	uint32_t n_jobs, order;
	uint64_t half, n_out;
};
hipError_t launch_hal_rows(hipStream_t s, int n_cu, const hal_rows_args &a);
hipError_t launch_hal_row(hipStream_t s, int n_cu, const void *evals, uint64_t len, f128 suffix, uint32_t order, uint64_t half, uint32_t point, f128 z,
                          void *out, uint64_t n_out);

// ---- kernels_misc.hip
hipError_t launch_inner_product(hipStream_t s, int n_cu, const void *a, uint32_t tower_level, const void *b,
                                uint64_t b_len, f128 *d_out);
hipError_t launch_fold_left(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec,
                            uint64_t vec_len, void *out, uint64_t out_len);
// fold_right as a GF(2)-linear map on the matrix cores (kernels_linmap.hip): rows of 512 / 1024 / 2048 bits; hipErrorNotSupported
// for every other shape.  d_table: linmap_table_bytes(row_bits) bytes of scratch.
size_t linmap_table_bytes(uint64_t row_bits);
hipError_t launch_fold_right_mfma(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec, uint64_t vec_len, void *out,
                                  uint64_t out_len, void *d_table);
hipError_t launch_fold_left_mfma(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec, uint64_t vec_len, void *out,
                                 uint64_t out_len, void *d_table); // B32 entries, vec_len 16 / 32 / 64
hipError_t launch_fold_right(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec,
                             uint64_t vec_len, void *out, uint64_t out_len);
hipError_t launch_compute_composite_generic(hipStream_t s, const void *const *d_rows_dev, uint32_t n_rows,
                                            uint64_t row_len, void *out, const bn_step *d_steps, uint32_t n_steps);
hipError_t launch_fri_fold(hipStream_t s, const uint64_t *d_s_evals, uint32_t tw_level, uint32_t log_domain,
                           uint32_t log_len, uint32_t log_batch, const f128 *h_challenges, uint32_t n_challenges,
                           const void *in, void *out, uint64_t out_len, void *scratch, int n_cu, const uint8_t *d_mul8);

// is eq[0 .. n) a tensor expansion up to a constant: eq[i] == eq[i - 2^k] * rho[k] (k = top bit of i)?  *d_flag |= 1 if not.
// d_rho[n_log] and d_first_wg[42] (from check_tensor_layout, which returns the grid size) are read by the kernel from memory.
uint32_t check_tensor_layout(uint32_t n_log, uint32_t *first_wg);
hipError_t launch_check_tensor(hipStream_t s, const void *eq, uint64_t n, const f128 *d_rho, const uint32_t *d_first_wg, uint32_t n_wg, uint32_t n_log,
                               unsigned *d_flag);

// ---- kernels_ntt.hip
hipError_t launch_ntt(hipStream_t s, bool inverse, void *data, uint32_t elem_level, uint32_t tw_level,
                      const uint64_t *d_s_evals, uint32_t log_domain, uint32_t log_x, uint32_t log_y, uint32_t log_z,
                      uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds);

// ---- kernels_mul9.hip: many element-wise products of equal length in ONE launch: out_j[i] = a_j[i] * b_j[i], i < n (the job table in
// pinned, device-mapped memory; the old HAL's wide zerocheck requests scale a column by the indicator table once per round)
struct mul9_job {
	const void *a, *b;
	void *out;
	const void *a2, *b2; // both null: out = a * b; both set: out = (a + a2) * (b + b2)
};
hipError_t launch_mul9_jobs(hipStream_t s, int n_cu, const mul9_job *d_jobs, uint32_t n_jobs, uint64_t n);
// ---- kernels_mul9.hip: out[i] = a[i*a_stride] * b[b_off + i*b_stride], bit-sliced
// up to four adjacent levels of pairwise_product_reduce in one launch of the element-wise product (kernels_mul9.hip:
// k_mul9_tree): level l (0-based) = products of adjacent pairs of lv[l - 1] (level 0: of `in`), n0 >> l of them, to lv[l]
struct mul9_tree_args {
	const uint32_t *in;
	uint32_t *lv[4];
	uint64_t n0;
	uint32_t n_levels;
};
hipError_t launch_mul9_tree(hipStream_t s, int n_cu, const mul9_tree_args &args);
// the small levels of pairwise_product_reduce (kernels_pairtree.hip): workgroup b walks n_levels <= log_s levels of the
// subtree over elements [b << n_levels, (b + 1) << n_levels) of `in`; level l (1-based) goes to out[l - 1]
struct pairtree_args {
	const f128 *in;
	f128 *out[8];
	uint32_t n_levels;
};
hipError_t launch_pairtree(hipStream_t s, const pairtree_args &args, uint32_t log_s, uint64_t n_groups);
hipError_t launch_mul9(hipStream_t s, int n_cu, const void *a, uint64_t a_stride, const void *b, uint64_t b_stride, uint64_t b_off,
                       void *out, uint64_t n);

// ---- kernels_ntt_tiled.hip
hipError_t launch_build_mul8(hipStream_t s, uint8_t *d_tab);
hipError_t launch_ntt_tiled(hipStream_t s, int n_cu, bool inverse, void *data, uint32_t elem_level, uint32_t tw_level,
                            const uint8_t *d_mul8, const uint64_t *d_s_evals, uint32_t log_domain, uint32_t log_x, uint32_t log_y,
                            uint32_t log_z, uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds);

hipError_t launch_ip32(hipStream_t s, int n_cu, const void *a_sub, const void *b, uint64_t n, f128 *d_out);
// device copy of the bit-sliced NTT's launch constants, kept across calls (kernels_ntt_bs.hip)
struct ntt_bs_cache {
	void *d_tables = nullptr;
	bool valid = false;
	uint64_t key = 0;
};
size_t ntt_bs_tables_bytes();
size_t ntt_bs_scratch_bytes(uint32_t log_words);
hipError_t launch_ntt_bs(hipStream_t s, bool inverse, void *data, const uint64_t *h_s_evals, uint32_t log_domain, uint32_t lx,
                         uint32_t log_y, uint32_t log_z, uint64_t coset, uint32_t coset_bits, uint32_t skip_rounds, void *d_scratch,
                         ntt_bs_cache *cache);


// ---- kernels_groestl.hip: Groestl-256 leaves, 2-to-1 compression layers, the flattened Merkle tree
hipError_t launch_groestl_leaves(hipStream_t s, int n_cu, const void *elems, uint64_t batch, uint64_t n_leaves, void *digests);
hipError_t launch_groestl_layer(hipStream_t s, int n_cu, const void *prev, uint64_t n_out, void *next);
hipError_t launch_gather(hipStream_t s, const void *src, const uint64_t *offsets, uint64_t n_items, uint64_t item_elems, void *out);
hipError_t launch_merkle_layers(hipStream_t s, int n_cu, void *nodes, uint64_t n_leaves);

} // namespace bn
