// tools/bsmul_rate.hip -- how fast can the SIMDs run the round-evaluation arithmetic when nothing
// but registers is involved?  One bs_mul<5> (+ accumulate) per iteration, optionally a 32x32
// transpose, at 1 and 2 workgroups of 256 threads per CU (= 1 and 2 waves per SIMD).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude tools/bsmul_rate.hip -o /tmp/bsmul_rate
#include <hip/hip_runtime.h>

#include <cstdio>

#include "bitslice.hpp"

using namespace bn;

template <bool TRANSPOSE, int WAVES, int SEQ = 0>
__global__ __launch_bounds__(256, WAVES) void k(uint32_t *out, int iters)
{
	uint32_t A[32], B[32], acc[32];
#pragma unroll
	for (int i = 0; i < 32; i++) {
		A[i] = threadIdx.x * 2654435761u + i * 40503u;
		B[i] = blockIdx.x * 2246822519u + i * 3266489917u + threadIdx.x;
		acc[i] = 0;
	}
	for (int it = 0; it < iters; it++) {
		uint32_t P[32];
		if (TRANSPOSE) transpose32(A);
		if constexpr (SEQ)
			bs_mul_seq<5, SEQ>(A, B, P);
		else
			bs_mul<5>(A, B, P);
#pragma unroll
		for (int i = 0; i < 32; i++) {
			acc[i] ^= P[i];
			A[i] ^= P[(i + 7) & 31]; // keep the next product dependent on this one
		}
	}
	uint32_t v = 0;
#pragma unroll
	for (int i = 0; i < 32; i++)
		v ^= acc[i] + A[i];
	out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <bool T, int W, int SEQ = 0>
static void run(const char *name, int blocks_per_cu)
{
	uint32_t *d;
	hipMalloc(&d, 256 * 2 * 256 * 4);
	const int iters = 2000;
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipLaunchKernelGGL((k<T, W, SEQ>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, 10);
	hipEventRecord(a);
	hipLaunchKernelGGL((k<T, W, SEQ>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, iters);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	// per SIMD: blocks_per_cu waves, each `iters` iterations
	printf("%-28s %d w/SIMD: %.3f ms, %.1f ns per iteration per wave, %.2f us per SIMD-iteration-pair\n", name, blocks_per_cu, ms,
	       ms * 1e6 / iters, ms * 1e3 / iters);
	hipFree(d);
}

int main()
{
	run<false, 2>("bs_mul<5>+acc", 1);
	run<false, 2>("bs_mul<5>+acc", 2);
	run<true, 2>("transpose32+bs_mul<5>+acc", 1);
	run<true, 2>("transpose32+bs_mul<5>+acc", 2);
	run<false, 1>("bs_mul<5>+acc (1-wave build)", 1);
	run<false, 2, 5>("bs_mul_seq<5,5>+acc", 1);
	run<false, 2, 5>("bs_mul_seq<5,5>+acc", 2);
	run<false, 2, 4>("bs_mul_seq<5,4>+acc", 1);
	run<false, 2, 4>("bs_mul_seq<5,4>+acc", 2);
	run<false, 2, 3>("bs_mul_seq<5,3>+acc", 1);
	run<false, 2, 3>("bs_mul_seq<5,3>+acc", 2);
	run<false, 3, 4>("bs_mul_seq<5,4>+acc (3-wave build)", 3);
	run<false, 3, 3>("bs_mul_seq<5,3>+acc (3-wave build)", 3);
	return 0;
}
