// binius_amd/csrc/finalize.hpp -- the tail of every accumulate_kernels launch: fold the raw device
// sums into the kernel's declared values (value = init + sum_t coeff_t * S[slot_t], the
// `*accumulator += ret * batch_coeff` of crates/compute/src/cpu/layer.rs:512), gather the returned
// values, re-zero the accumulator slots and publish to the host mailbox.
//
// Shared by the stand-alone k_finalize kernel and by the LAST workgroup of the round-eval kernel
// (fused form: one launch per round evaluation instead of two).
#pragma once
#include <hip/hip_runtime.h>

#include "gf128.hpp"
#include "internal.hpp"

// tools/small_round_phases.hip defines BN_PHASE_TS: thread 0 of workgroup 0 stamps the 100 MHz wall clock at the
// phase boundaries of a small round (never defined in the library build)
#ifdef BN_PHASE_TS
__device__ uint64_t bn_phase_ts[16];
#define BN_TS(i)                                                             \
	do {                                                                     \
		if (blockIdx.x == 0 && threadIdx.x == 0) ::bn_phase_ts[i] = wall_clock64(); \
	} while (0)
#else
#define BN_TS(i)
#endif
#define BN_FTS(i) BN_TS(i)

namespace bn {

__device__ __forceinline__ uint32_t fin_wave_xor(uint32_t v)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	return v;
}

// ---- cross-rank reduction (fin_peer, internal.hpp) -------------------------------------------------------------------
// Wave 0 of the finalizing workgroup; vals[0 .. n_ret) in LDS hold this rank's returned values on entry and the XOR over
// all ranks on exit.  Lane p < world talks to rank p: it stores the values into slot `rank` of p's mailbox (system-scope
// stores into fine-grained memory), then the slot's TAG word, and waits for slot p of its OWN mailbox to become valid for
// this round before it takes rank p's values.
//
// Slots validate themselves (the ordering of the value words and the flag must not be an argument about the
// fabric): tag = peer_tag(round, value words), a 64-bit mix of the round number and every value word.  The reader loads
// the tag AND the values in one look and accepts them only if the tag it read is the tag OF the values it read for THIS
// round; a tag that overtook a value word over xGMI, a torn slot or a stale one (same parity, round - 2) does not
// validate and the lane looks again.  So a stale partial can never be XORed into a round polynomial silently, whatever
// order the words arrive in; the writer still drains its value stores (vmcnt(0)) before the tag so that the first look
// after the tag lands normally succeeds.  Two parities: a rank can be at most one round ahead of the slowest reader (it
// needs everybody's round r + 1 values, which they send after reading round r).  The wait is bounded (a rank that
// died cannot park the device): on a timeout the function returns false and the caller reports it through mail[65].
//
// stress (BN_PEER_STRESS, test only): bit 0 = the tag is stored FIRST and the values follow after a pause -- the order a
// reordering fabric could produce; bit 1 = a pause between the value words as well (torn slots).
__device__ __forceinline__ uint64_t peer_mix(uint64_t h, uint64_t w)
{
	h = (h ^ w) * 0xBF58476D1CE4E5B9ull;
	return h ^ (h >> 29);
}
__device__ __forceinline__ uint64_t peer_tag(uint64_t round, const uint64_t *w, uint32_t n_words)
{
	uint64_t h = peer_mix(0x9E3779B97F4A7C15ull, round);
	for (uint32_t i = 0; i < n_words; i++) h = peer_mix(h, w[i]);
	return h | 1ull; // (never the zero a fresh mailbox holds)
}

__device__ __forceinline__ bool peer_exchange(f128 *vals, uint32_t n_ret, uint64_t *const *box, uint32_t world, uint32_t rank, uint64_t round,
                                              uint32_t stress = 0)
{
	const unsigned p = threadIdx.x; // < 64
	const unsigned par = (unsigned)(round & 1);
	bool ok = true;
	uint64_t got[2 * kFinMaxRets];
#pragma unroll
	for (int i = 0; i < 2 * kFinMaxRets; i++) got[i] = 0;
	if (p < world) {
		uint64_t mine[2 * kFinMaxRets];
#pragma unroll
		for (int r = 0; r < kFinMaxRets; r++) {
			mine[2 * r] = (uint32_t)r < n_ret ? vals[r].lo : 0;
			mine[2 * r + 1] = (uint32_t)r < n_ret ? vals[r].hi : 0;
		}
		const uint64_t tag = peer_tag(round, mine, 2 * n_ret);
		uint64_t *dst = box[p] + (size_t)(par * kPeerMaxWorld + rank) * kPeerSlotWords;
		if (stress & 1) { // the flag overtakes the values
			__hip_atomic_store(dst + 16, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			for (int k = 0; k < 64; k++) __builtin_amdgcn_s_sleep(127);
		}
#pragma unroll
		for (int r = 0; r < kFinMaxRets; r++)
			if ((uint32_t)r < n_ret) {
				__hip_atomic_store(dst + 2 * r, mine[2 * r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				if (stress & 2) {
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					for (int k = 0; k < 8; k++) __builtin_amdgcn_s_sleep(127);
				}
				__hip_atomic_store(dst + 2 * r + 1, mine[2 * r + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		// No release / acquire FENCES: a system-scope release writes the whole L2 back and an acquire invalidates it (measured:
		// +6 us per round on one device).  Every word of the mailbox is only ever touched by system-scope atomics on
		// fine-grained memory, performed at the memory itself; the drain below makes the common case cheap (the tag lands after
		// the values), the tag's content makes every other case safe.
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (!(stress & 1)) __hip_atomic_store(dst + 16, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		const uint64_t *src = box[rank] + (size_t)(par * kPeerMaxWorld + p) * kPeerSlotWords;
		uint32_t spins = 0;
		for (;;) {
			const uint64_t t = __hip_atomic_load(src + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
			for (int r = 0; r < kFinMaxRets; r++)
				if ((uint32_t)r < n_ret) {
					got[2 * r] = __hip_atomic_load(src + 2 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					got[2 * r + 1] = __hip_atomic_load(src + 2 * r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			if (t == peer_tag(round, got, 2 * n_ret)) break; // this round's values, all of them
			if (++spins > (1u << 21)) { // ~1 us per look: gives up after a couple of seconds
				ok = false;
				break;
			}
			__builtin_amdgcn_s_sleep(1);
		}
	}
	ok = __all(ok);
#pragma unroll
	for (int r = 0; r < kFinMaxRets; r++) {
		if ((uint32_t)r >= n_ret) break;
		uint64_t lo = (p < world && ok) ? got[2 * r] : 0, hi = (p < world && ok) ? got[2 * r + 1] : 0;
#pragma unroll
		for (int m = 8; m >= 1; m >>= 1) { // world <= 16
			lo ^= __shfl_xor(lo, m, 64);
			hi ^= __shfl_xor(hi, m, 64);
		}
		if (p == 0) vals[r] = f128{lo, hi};
	}
	return ok;
}

// Must be called by EVERY thread of a workgroup with >= 128 threads (it contains barriers).
// S_local == nullptr: the raw sums are in global memory (S, XOR-accumulated by all workgroups) and are
// read with agent-scope atomic loads, all terms at once (one memory round trip), then re-zeroed.
// S_local != nullptr: a single-workgroup launch kept its sums in LDS (S_local[slot]); global S was
// never touched.
// Term t is folded in by 128 threads (lane i contributes bit_i(S) ? coeff * 2^i : 0, gf128.hpp
// mul_basis); a workgroup of 256 threads handles two terms per pass.
__device__ __forceinline__ void finalize_body(const fin_args &a, f128 *S, f128 *rets, f128 *mail, uint64_t seq,
                                              const f128 *S_local = nullptr, const fin_peer *peer = nullptr)
{
	__shared__ f128 fin_S[kFinMaxTerms];
	__shared__ uint64_t fin_red[4][2][2];   // [term group][wave in group][lo/hi]
	__shared__ f128 fin_values[kFinMaxValues];
	const unsigned tid = threadIdx.x;
	const unsigned n_groups = blockDim.x >= 512 ? 4u : (blockDim.x >= 256 ? 2u : 1u);
	if (tid < a.n_values)
		fin_values[tid] = a.init[tid];
	if (tid < a.n_terms) {
		const uint32_t slot = a.terms[tid].slot;
		f128 s;
		if (S_local) {
			s = S_local[slot];
		} else {
			s.lo = __hip_atomic_load(&S[slot].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s.hi = __hip_atomic_load(&S[slot].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		fin_S[tid] = s;
	}
	__syncthreads();
	BN_FTS(9);
	const unsigned grp = tid >> 7, l128 = tid & 127;
	for (uint32_t t0 = 0; t0 < a.n_terms; t0 += n_groups) {
		const uint32_t t = t0 + grp;
		f128 c = f128_zero();
		if (grp < n_groups && t < a.n_terms) {
			const f128 s = fin_S[t];
			const f128 coeff = a.terms[t].coeff;
			if (coeff.lo == 1 && coeff.hi == 0) {
				// batch coefficient alpha^0 = 1 (the only one on the measured single-claim path)
				if (l128 == 0) c = s;
			} else {
				const uint64_t word = l128 < 64 ? s.lo : s.hi;
				if ((word >> (l128 & 63)) & 1)
					c = mul_basis(coeff, l128);
			}
		}
		uint32_t w[4] = {(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32)};
#pragma unroll
		for (int q = 0; q < 4; q++)
			w[q] = fin_wave_xor(w[q]);
		if (grp < n_groups && (tid & 63) == 0) {
			fin_red[grp][l128 >> 6][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
			fin_red[grp][l128 >> 6][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
		}
		__syncthreads();
		if (tid == 0) {
			for (unsigned gq = 0; gq < n_groups && t0 + gq < a.n_terms; gq++) {
				const uint32_t v = a.terms[t0 + gq].value;
				fin_values[v].lo ^= fin_red[gq][0][0] ^ fin_red[gq][1][0];
				fin_values[v].hi ^= fin_red[gq][0][1] ^ fin_red[gq][1][1];
			}
		}
		__syncthreads();
	}
	BN_FTS(10);
	if (peer && peer->world > 1) {
		// returned values of this rank -> XOR over all ranks (the barriers are uniform: peer is a kernel argument)
		__shared__ f128 fin_pv[kFinMaxRets];
		if (tid < a.n_ret) fin_pv[tid] = fin_values[a.ret_ids[tid]];
		__syncthreads();
		if (tid < 64) {
			const bool ok = peer_exchange(fin_pv, a.n_ret, peer->box, peer->world, peer->rank, peer->round, peer->stress);
			if (!ok && tid == 0 && seq) __hip_atomic_store(&mail[65].lo, peer->round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		__syncthreads();
		if (tid < a.n_ret) fin_values[a.ret_ids[tid]] = fin_pv[tid];
		__syncthreads();
	}
	if (tid < a.n_ret) {
		const f128 v = fin_values[a.ret_ids[tid]];
		rets[tid] = v;
		if (seq) {
			// zero-copy return: values, then the sequence word, into fine-grained host memory
			__hip_atomic_store(&mail[tid].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(&mail[tid].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	// leave the accumulator slots zero for the next launch (no memset on the per-round path)
	if (!S_local && tid < a.n_slots) {
		__hip_atomic_store(&S[tid].lo, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(&S[tid].hi, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	BN_FTS(11);
	if (seq) {
		// n_ret <= 8: the value stores above were issued by lanes of wave 0 (system-scope atomics into fine-grained host
		// memory: they do not sit in the L2); the wave drains them (vmcnt(0)) and then writes the sequence word -- posted
		// writes to one destination keep their order.  (A release here would write the whole L2 back first.)
		if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (tid == 0)
			__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// ---- finalize arguments staged in LDS -----------------------------------------------------------------------------
// fin_fuse travels as a kernel argument (1.3 KiB).  Read where it is needed -- at the very end of the kernel -- every
// field is a fresh round trip to the kernarg segment, and they depend on one another (n_terms -> terms[t].slot ->
// S[slot] -> terms[t].coeff -> ...): ~2.3 us of a 13 us small round (tools/small_round_phases.hip).  The latency-shaped
// kernels therefore issue all those loads at kernel entry (fin_prefetch), park them in LDS once the first barrier has
// passed anyway (fin_commit) and finalize from there (finalize_cached).
struct fin_cache {
	f128 coeff[kFinMaxTerms];
	f128 init[kFinMaxValues];
	uint32_t slot[kFinMaxTerms], value[kFinMaxTerms];
	uint32_t ret_ids[kFinMaxRets];
	uint32_t n_terms, n_values, n_ret, n_slots;
	uint32_t all_one; // every batch coefficient is 1: value = init ^ XOR of its sums, no multiplication
	f128 *S, *rets, *mail;
	unsigned *counter;
	uint64_t *peer_box[kPeerMaxWorld];
	uint32_t peer_world, peer_rank, peer_stress;
	uint64_t peer_round;
};
struct fin_pref {
	f128 coeff, init;
	uint32_t slot, value, ret_id;
	uint64_t *peer_box;
};

// every thread of the workgroup (>= 64 threads); only the first kFinMaxTerms lanes load
__device__ __forceinline__ fin_pref fin_prefetch(const fin_fuse &fz)
{
	fin_pref r{};
	const unsigned tid = threadIdx.x;
	if (tid < kFinMaxTerms) {
		r.coeff = fz.args.terms[tid].coeff;
		r.slot = fz.args.terms[tid].slot;
		r.value = fz.args.terms[tid].value;
		r.init = fz.args.init[tid & (kFinMaxValues - 1)];
		r.ret_id = fz.args.ret_ids[tid & (kFinMaxRets - 1)];
		r.peer_box = fz.peer.box[tid & (kPeerMaxWorld - 1)];
	}
	return r;
}

// a workgroup barrier must separate this from finalize_cached (every kernel has several)
__device__ __forceinline__ void fin_commit(const fin_fuse &fz, const fin_pref &r, fin_cache &c)
{
	const unsigned tid = threadIdx.x;
	if (tid < 64) {
		const uint32_t n_terms = fz.args.n_terms;
		const bool one = tid >= n_terms || (r.coeff.lo == 1 && r.coeff.hi == 0);
		const bool all = __all(one);
		if (tid < kFinMaxTerms) {
			c.coeff[tid] = r.coeff;
			c.slot[tid] = r.slot;
			c.value[tid] = r.value;
		}
		if (tid < kFinMaxValues) c.init[tid] = r.init;
		if (tid < kFinMaxRets) c.ret_ids[tid] = r.ret_id;
		if (tid < kPeerMaxWorld) c.peer_box[tid] = r.peer_box;
		if (tid == 0) {
			c.peer_world = fz.peer.world;
			c.peer_rank = fz.peer.rank;
			c.peer_stress = fz.peer.stress;
			c.peer_round = fz.peer.round;
			c.n_terms = n_terms;
			c.n_values = fz.args.n_values;
			c.n_ret = fz.args.n_ret;
			c.n_slots = fz.args.n_slots;
			c.all_one = all ? 1u : 0u;
			c.S = fz.S;
			c.rets = fz.rets;
			c.mail = fz.mail;
			c.counter = fz.counter;
		}
	}
}

// finalize_body with every argument taken from LDS.  Same contract: all threads of a workgroup of >= 128 threads.
__device__ __forceinline__ void finalize_cached(const fin_cache &c, uint64_t seq, const f128 *S_local = nullptr)
{
	__shared__ f128 fc_S[kFinMaxTerms];
	__shared__ uint64_t fc_red[4][2][2];
	__shared__ f128 fc_values[kFinMaxValues];
	const unsigned tid = threadIdx.x;
	const uint32_t n_terms = c.n_terms, n_values = c.n_values, n_ret = c.n_ret;
	f128 *S = c.S, *rets = c.rets, *mail = c.mail;
	if (tid < n_terms) {
		const uint32_t slot = c.slot[tid];
		f128 s;
		if (S_local) {
			s = S_local[slot];
		} else {
			s.lo = __hip_atomic_load(&S[slot].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s.hi = __hip_atomic_load(&S[slot].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		fc_S[tid] = s;
	}
	__syncthreads();
	BN_FTS(9);
	if (c.all_one) {
		// the single-claim path: no batch coefficient to multiply by
		if (tid < n_values) {
			f128 v = c.init[tid];
			for (uint32_t t = 0; t < n_terms; t++)
				if (c.value[t] == tid) v ^= fc_S[t];
			fc_values[tid] = v;
		}
		__syncthreads();
	} else {
		const unsigned n_groups = blockDim.x >= 512 ? 4u : (blockDim.x >= 256 ? 2u : 1u);
		if (tid < n_values) fc_values[tid] = c.init[tid];
		const unsigned grp = tid >> 7, l128 = tid & 127;
		for (uint32_t t0 = 0; t0 < n_terms; t0 += n_groups) {
			const uint32_t t = t0 + grp;
			f128 cf = f128_zero();
			if (grp < n_groups && t < n_terms) {
				const f128 s = fc_S[t];
				const f128 coeff = c.coeff[t];
				if (coeff.lo == 1 && coeff.hi == 0) {
					if (l128 == 0) cf = s;
				} else {
					const uint64_t word = l128 < 64 ? s.lo : s.hi;
					if ((word >> (l128 & 63)) & 1) cf = mul_basis(coeff, l128);
				}
			}
			uint32_t w[4] = {(uint32_t)cf.lo, (uint32_t)(cf.lo >> 32), (uint32_t)cf.hi, (uint32_t)(cf.hi >> 32)};
#pragma unroll
			for (int q = 0; q < 4; q++)
				w[q] = fin_wave_xor(w[q]);
			if (grp < n_groups && (tid & 63) == 0) {
				fc_red[grp][l128 >> 6][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
				fc_red[grp][l128 >> 6][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
			}
			__syncthreads();
			if (tid == 0) {
				for (unsigned gq = 0; gq < n_groups && t0 + gq < n_terms; gq++) {
					const uint32_t v = c.value[t0 + gq];
					fc_values[v].lo ^= fc_red[gq][0][0] ^ fc_red[gq][1][0];
					fc_values[v].hi ^= fc_red[gq][0][1] ^ fc_red[gq][1][1];
				}
			}
			__syncthreads();
		}
	}
	BN_FTS(10);
	if (c.peer_world > 1) {
		__shared__ f128 fc_pv[kFinMaxRets];
		if (tid < n_ret) fc_pv[tid] = fc_values[c.ret_ids[tid]];
		__syncthreads();
		if (tid < 64) {
			const bool ok = peer_exchange(fc_pv, n_ret, c.peer_box, c.peer_world, c.peer_rank, c.peer_round, c.peer_stress);
			if (!ok && tid == 0 && seq) __hip_atomic_store(&mail[65].lo, c.peer_round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		__syncthreads();
		if (tid < n_ret) fc_values[c.ret_ids[tid]] = fc_pv[tid];
		__syncthreads();
	}
	if (tid < n_ret) {
		const f128 v = fc_values[c.ret_ids[tid]];
		rets[tid] = v;
		if (seq) {
			__hip_atomic_store(&mail[tid].lo, v.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(&mail[tid].hi, v.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	if (!S_local && tid < c.n_slots) {
		__hip_atomic_store(&S[tid].lo, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(&S[tid].hi, (uint64_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	BN_FTS(11);
	if (seq) {
		if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (as in finalize_body: no release fence)
		if (tid == 0)
			__hip_atomic_store(&mail[64].lo, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

} // namespace bn
