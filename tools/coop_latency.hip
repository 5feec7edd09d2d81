// tools/coop_latency.hip -- what would a grid-wide barrier inside the claim-group launch cost?  (a parallel
// fold phase and a parallel evaluate phase of ONE launch instead of a chain.)
//  (1) a 256 x 768-thread kernel (the group kernel's shape: one workgroup per CU) that publishes a sequence word, launched the
//      ordinary way, host spinning on the word;
//  (2) the same through hipLaunchCooperativeKernel;
//  (3) the cooperative kernel with K grid-wide barriers (cooperative_groups::grid_group::sync) in front of the publish: the
//      difference per barrier;
//  (4) the same K barriers hand-made (one device-scope counter, every workgroup's thread 0 adds and spins; only valid because the
//      grid is co-resident) in the ORDINARY launch -- what a persistent-grid barrier costs without the cooperative API.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coop_latency.hip -o tools/coop_latency
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

namespace cg = cooperative_groups;

__global__ __launch_bounds__(768) void k_plain(volatile uint64_t *mail, uint64_t seq, unsigned *counter, int barriers, unsigned base)
{
	for (int b = 1; b <= barriers; b++) {
		__syncthreads();
		if (threadIdx.x == 0) {
			__threadfence();
			atomicAdd(counter, 1u);
			const unsigned want = base + (unsigned)b * gridDim.x; // (the counter only ever grows: no reset to race with)
			unsigned spins = 0;
			while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
				if (++spins > (1u << 26)) break; // bounded
				__builtin_amdgcn_s_sleep(1);
			}
		}
		__syncthreads();
	}
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		__hip_atomic_store((uint64_t *)mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

__global__ __launch_bounds__(768) void k_coop(volatile uint64_t *mail, uint64_t seq, int barriers)
{
	cg::grid_group g = cg::this_grid();
	for (int b = 0; b < barriers; b++) g.sync();
	if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store((uint64_t *)mail, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main()
{
	uint64_t *h = nullptr, *d = nullptr;
	hipHostMalloc((void **)&h, 4096, hipHostMallocMapped | hipHostMallocCoherent);
	hipHostGetDevicePointer((void **)&d, h, 0);
	unsigned *d_counter = nullptr;
	hipMalloc((void **)&d_counter, 64);
	hipMemset(d_counter, 0, 64);
	hipStream_t s;
	hipStreamCreate(&s);
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const unsigned grid = (unsigned)prop.multiProcessorCount;
	printf("grid %u x 768 threads, cooperativeLaunch = %d\n", grid, prop.cooperativeLaunch);
	const int N = 1000;
	uint64_t seq = 0;
	auto wait = [&](uint64_t v) {
		uint64_t spins = 0;
		while (__atomic_load_n(&h[0], __ATOMIC_ACQUIRE) != v)
			if (++spins > (1ull << 31)) {
				printf("timeout\n");
				return false;
			}
		return true;
	};
	unsigned base = 0;
	for (int barriers : {0, 1, 4}) {
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < N; i++) {
			++seq;
			hipLaunchKernelGGL(k_plain, dim3(grid), dim3(768), 0, s, d, seq, d_counter, barriers, base);
			base += (unsigned)barriers * grid;
			if (!wait(seq)) return 1;
		}
		auto t1 = std::chrono::steady_clock::now();
		printf("ordinary launch, %d hand-made grid barrier(s): %.2f us per round trip\n", barriers, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
	}
	for (int barriers : {0, 1, 4}) {
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < N; i++) {
			++seq;
			uint64_t sq = seq;
			int bb = barriers;
			void *args[] = {(void *)&d, (void *)&sq, (void *)&bb};
			const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&k_coop), dim3(grid), dim3(768), args, 0, s);
			if (e != hipSuccess) {
				printf("hipLaunchCooperativeKernel: %s\n", hipGetErrorString(e));
				return 1;
			}
			if (!wait(seq)) return 1;
		}
		auto t1 = std::chrono::steady_clock::now();
		printf("cooperative launch, %d grid.sync(): %.2f us per round trip\n", barriers, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
	}
	hipStreamSynchronize(s);
	return 0;
}
