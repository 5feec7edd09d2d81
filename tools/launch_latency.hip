// tools/launch_latency.hip -- how long is one host <-> device round trip on this box?
//  (1) launch: host launches a trivial kernel that publishes a sequence word into fine-grained
//      pinned host memory; the host spins on it, then launches the next one.
//  (2) persistent: ONE kernel stays resident and ping-pongs with the host through two words of
//      pinned host memory (host writes `go`, kernel answers `seq`), bounded spin.
// Build: hipcc -O3 --offload-arch=gfx950 tools/launch_latency.hip -o /tmp/launch_latency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void k_reply(volatile uint64_t *mail, uint64_t seq)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		__hip_atomic_store((uint64_t *)mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

__global__ void k_persistent(volatile uint64_t *go, volatile uint64_t *mail, uint64_t rounds)
{
	if (threadIdx.x != 0) return;
	for (uint64_t r = 1; r <= rounds; r++) {
		uint64_t spins = 0;
		while (__hip_atomic_load((uint64_t *)go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < r) {
			if (++spins > (1ull << 24)) return; // bounded
			__builtin_amdgcn_s_sleep(1);
		}
		__hip_atomic_store((uint64_t *)mail, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

int main()
{
	uint64_t *h = nullptr, *d = nullptr;
	hipHostMalloc((void **)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent);
	hipHostGetDevicePointer((void **)&d, h, 0);
	h[0] = 0;
	h[8] = 0;
	hipStream_t s;
	hipStreamCreate(&s);
	const int N = 2000;
	for (int rep = 0; rep < 2; rep++) {
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 1; i <= N; i++) {
			const uint64_t seq = (uint64_t)rep * N + i;
			hipLaunchKernelGGL(k_reply, dim3(1), dim3(64), 0, s, d, seq);
			while (__atomic_load_n(&h[0], __ATOMIC_ACQUIRE) != seq) {
			}
		}
		auto t1 = std::chrono::steady_clock::now();
		printf("launch ping-pong      : %.2f us per round trip\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
	}
	// two dependent launches per host wait (the unfused fold + eval pattern)
	{
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 1; i <= N; i++) {
			const uint64_t seq = 100000 + i;
			hipLaunchKernelGGL(k_reply, dim3(1), dim3(64), 0, s, d + 16, seq);
			hipLaunchKernelGGL(k_reply, dim3(1), dim3(64), 0, s, d, seq);
			while (__atomic_load_n(&h[0], __ATOMIC_ACQUIRE) != seq) {
			}
		}
		auto t1 = std::chrono::steady_clock::now();
		printf("2 launches + wait     : %.2f us per round trip\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
	}
	hipStreamSynchronize(s);
	h[0] = 0;
	h[8] = 0;
	hipLaunchKernelGGL(k_persistent, dim3(1), dim3(64), 0, s, d + 8, d, (uint64_t)N);
	auto t0 = std::chrono::steady_clock::now();
	for (int i = 1; i <= N; i++) {
		__atomic_store_n(&h[8], (uint64_t)i, __ATOMIC_RELEASE);
		uint64_t spins = 0;
		while (__atomic_load_n(&h[0], __ATOMIC_ACQUIRE) != (uint64_t)i) {
			if (++spins > (1ull << 30)) {
				printf("persistent: timeout at %d\n", i);
				return 1;
			}
		}
	}
	auto t1 = std::chrono::steady_clock::now();
	printf("persistent ping-pong  : %.2f us per round trip\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
	hipStreamSynchronize(s);
	return 0;
}
