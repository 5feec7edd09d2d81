// binius_amd/csrc/arm.hpp -- "armed" rounds: a small-round kernel that is already on the device when its challenge
// arrives.
//
// A small sumcheck round costs ~10 us of kernel and ~7 us of launch (runtime + command processor + wave start,
// tools/small_round_phases.hip), and the launch cannot start before the host knows the challenge.  But everything else
// about the next round is known when the current one is launched: the arrays (the folded halves, in place), the size,
// the finalize recipe, the mailbox sequence number.  So right behind the kernel of round r the dispatcher (abi_kernels.cpp)
// enqueues the kernel of round r + 1 "armed": it starts as soon as round r has finished, issues its loads, and waits for
// the one thing it lacks -- z -- on a command word in pinned host memory.  When the caller's next fold + evaluation
// request turns out to be the predicted one, the host writes (z, command) instead of launching: ~2 us instead of ~7.
// Anything else (a different call, other arrays, a different recipe) cancels the kernel, which then exits without
// having touched anything; so does a bounded spin, so a host that went away cannot hang the device.
//
// Protocol.  Ids increase by one per armed kernel.  Command word (host -> device, h_cmd[0]) = (id << 2) | code with
// code 1 = go, 2 = cancel; h_cmd[2..3] = z, h_cmd[4..5] = hi_scale, written before the word.  A kernel that reads a
// word with a larger id than its own was cancelled (the host only moves forward).  Only workgroup 0 polls the host;
// it republishes (word, z, hi_scale) in device memory (d_relay) for the others.  On a timeout workgroup 0 relays a
// cancel and writes h_cmd[6] = id (status: "left without running"), which the host checks while it waits for the
// mailbox.
#pragma once
#include <hip/hip_runtime.h>

#include "gf128.hpp"

namespace bn {

struct arm_args {
	const uint64_t *h_cmd; // pinned host memory (device view); nullptr = a normal launch
	uint64_t *h_status;    // &h_cmd[6]
	uint64_t *d_relay;     // device memory, 8 words
	uint64_t id;
};

constexpr uint64_t kArmGo = 1, kArmCancel = 2;
constexpr uint64_t kArmLost = 1ull << 63; // status word: a workgroup other than 0 gave up waiting for the relay

// All threads of the workgroup.  Returns true (and z, hi_scale) when the round is to run, false when the kernel has
// to leave.  Contains one workgroup barrier.
__device__ __forceinline__ bool arm_wait(const arm_args &arm, f128 &z, f128 &hi_scale)
{
	__shared__ uint64_t arm_box[5];
	if (threadIdx.x == 0) {
		uint64_t code = kArmCancel;
		uint64_t zl = 0, zh = 0, hl = 0, hh = 0;
		if (blockIdx.x == 0) {
			bool timed_out = true;
			for (uint32_t spins = 0; spins < (1u << 12); spins++) { // ~1.4 us per poll: gives up after ~6 ms
				const uint64_t w = __hip_atomic_load(arm.h_cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
				if ((w >> 2) >= arm.id) {
					timed_out = false;
					if (w == ((arm.id << 2) | kArmGo)) {
						code = kArmGo;
						zl = __hip_atomic_load(arm.h_cmd + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						zh = __hip_atomic_load(arm.h_cmd + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						hl = __hip_atomic_load(arm.h_cmd + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						hh = __hip_atomic_load(arm.h_cmd + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					}
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			if (timed_out) __hip_atomic_store(arm.h_status, arm.id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (gridDim.x > 1) {
				__hip_atomic_store(arm.d_relay + 2, zl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(arm.d_relay + 3, zh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(arm.d_relay + 4, hl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(arm.d_relay + 5, hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				// (device-scope atomics are performed at the coherence point in program order per lane: s_waitcnt orders them)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__hip_atomic_store(arm.d_relay, (arm.id << 2) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		} else {
			// Workgroup 0's bounded spin on the host (~6 ms) is the only DECISION; everybody else just waits for its relay.
			// The bound here (seconds) only exists so that a lost workgroup 0 cannot park the device for ever; a
			// workgroup that ever runs into it says so in the status word (kArmLost): it leaves without a ticket, so the
			// round's result is never published and the host, which checks the status word while it waits for the
			// mailbox, reports a device error instead of accepting anything from a partially executed round.
			bool lost = true;
			for (uint32_t spins = 0; spins < (1u << 22); spins++) {
				const uint64_t w = __hip_atomic_load(arm.d_relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((w >> 2) == arm.id) {
					code = w & 3;
					lost = false;
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			if (lost) __hip_atomic_store(arm.h_status, arm.id | kArmLost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (code == kArmGo) {
				zl = __hip_atomic_load(arm.d_relay + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				zh = __hip_atomic_load(arm.d_relay + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				hl = __hip_atomic_load(arm.d_relay + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				hh = __hip_atomic_load(arm.d_relay + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		arm_box[0] = code;
		arm_box[1] = zl;
		arm_box[2] = zh;
		arm_box[3] = hl;
		arm_box[4] = hh;
	}
	__syncthreads();
	const volatile uint32_t *bw = reinterpret_cast<const volatile uint32_t *>(arm_box);
	// (readfirstlane: keeps z wave-uniform = in SGPRs, like the kernel argument it replaces)
	const uint32_t c0 = __builtin_amdgcn_readfirstlane(bw[0]);
	uint32_t w[8];
#pragma unroll
	for (int i = 0; i < 8; i++)
		w[i] = __builtin_amdgcn_readfirstlane(bw[2 + i]);
	z = f128{(uint64_t)w[0] | ((uint64_t)w[1] << 32), (uint64_t)w[2] | ((uint64_t)w[3] << 32)};
	hi_scale = f128{(uint64_t)w[4] | ((uint64_t)w[5] << 32), (uint64_t)w[6] | ((uint64_t)w[7] << 32)};
	return c0 == kArmGo;
}

} // namespace bn
