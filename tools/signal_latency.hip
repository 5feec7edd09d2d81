// tools/signal_latency.hip -- host -> device -> host signalling round trip for the armed rounds (csrc/arm.hpp):
// where should the command word live?  (a) pinned host memory polled by the device over PCIe (what arm.hpp does),
// (b) device memory the host writes through the BAR (fine-grained hipExtMallocWithFlags, or plain hipMalloc if the
// platform maps it), polled locally.  The ack always goes to pinned host memory.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/signal_latency.hip -o tools/signal_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <vector>

__global__ void k_pingpong(const uint64_t *cmd, uint64_t *ack, int rounds, int system_scope)
{
	for (int r = 1; r <= rounds; r++) {
		for (uint32_t spins = 0; spins < (1u << 24); spins++) {
			const uint64_t w = system_scope ? __hip_atomic_load(cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
			                                : __hip_atomic_load(cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (w >= (uint64_t)r) break;
			__builtin_amdgcn_s_sleep(1);
		}
		__hip_atomic_store(ack, (uint64_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static void run(const char *name, uint64_t *cmd_host_view, const uint64_t *cmd_dev_view)
{
	uint64_t *ack;
	(void)hipHostMalloc(&ack, 64, hipHostMallocMapped | hipHostMallocCoherent);
	*ack = 0;
	signal(SIGSEGV, on_segv);
	signal(SIGBUS, on_segv);
	if (sigsetjmp(jb, 1)) {
		printf("%-44s host write faults: not host-accessible\n", name);
		return;
	}
	*(volatile uint64_t *)cmd_host_view = 0;
	const int rounds = 2000;
	hipStream_t s;
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, s, cmd_dev_view, ack, rounds, 1);
	std::vector<double> us;
	for (int r = 1; r <= rounds; r++) {
		const auto t0 = std::chrono::steady_clock::now();
		__atomic_store_n(cmd_host_view, (uint64_t)r, __ATOMIC_RELEASE);
		uint64_t spins = 0;
		while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != (uint64_t)r)
			if (++spins > (1ull << 28)) {
				printf("%-44s no answer (the device does not see the host's writes)\n", name);
				__atomic_store_n(cmd_host_view, (uint64_t)rounds + 1, __ATOMIC_RELEASE);
				(void)hipStreamSynchronize(s);
				return;
			}
		const auto t1 = std::chrono::steady_clock::now();
		if (r > 100) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
	}
	(void)hipStreamSynchronize(s);
	std::sort(us.begin(), us.end());
	printf("%-44s round trip median %.2f us, p10 %.2f, p90 %.2f\n", name, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
}

int main()
{
	uint64_t *h, *d_h;
	(void)hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent);
	(void)hipHostGetDevicePointer((void **)&d_h, h, 0);
	run("pinned host memory (device polls over PCIe)", h, d_h);
	uint64_t *fg = nullptr;
	if (hipExtMallocWithFlags((void **)&fg, 64, hipDeviceMallocFinegrained) == hipSuccess && fg)
		run("fine-grained device memory (host writes BAR)", fg, fg);
	else
		printf("hipExtMallocWithFlags(finegrained) failed\n");
	uint64_t *dm = nullptr;
	if (hipMalloc((void **)&dm, 64) == hipSuccess) run("plain hipMalloc (host writes BAR)", dm, dm);
	uint64_t *mg = nullptr;
	if (hipMallocManaged((void **)&mg, 64) == hipSuccess) {
		(void)hipMemAdvise(mg, 64, hipMemAdviseSetPreferredLocation, 0);
		run("managed memory, preferred on device", mg, mg);
	}
	return 0;
}
