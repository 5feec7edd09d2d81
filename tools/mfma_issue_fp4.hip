// tools/mfma_issue_fp4.hip -- tools/mfma_issue.hip for the FP4 matrix path: how do v_mfma_scale_f32_32x32x64_f8f6f4 (both operands
// E2M1) and the bitwise VALU that prepares its operands share a SIMD?  Nine independent accumulator tiles per iteration (CHAIN = 0)
// or one dependent chain (CHAIN = 1: every MFMA accumulates into the same tile, the shape of kernels_linmap.hip); V bitwise VALU
// instructions per MFMA (4 of them the operand masks).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_issue_fp4.hip -o tools/mfma_issue_fp4
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define MFMA4(ACC, A, B)                                                                                                                       \
	ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{(A).x, (A).y, (A).z, (A).w, 0, 0, 0, 0}, v8i{(B).x, (B).y, (B).z, (B).w, 0, 0, 0, 0}, ACC, \
	                                                      4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F)

__device__ __forceinline__ v4i and4(v4i x, uint32_t m)
{
	return v4i{(int)((uint32_t)x.x & m), (int)((uint32_t)x.y & m), (int)((uint32_t)x.z & m), (int)((uint32_t)x.w & m)};
}

template <int CHAIN, int EXTRA, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k(uint32_t *out, int iters)
{
	v16f acc[9];
#pragma unroll
	for (int i = 0; i < 9; i++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[i][r] = 0.0f;
	v4i u[4], v[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		u[j] = v4i{(int)(threadIdx.x * 2654435761u + j), (int)(threadIdx.x * 40503u + j), (int)(blockIdx.x + j), (int)(j * 77 + threadIdx.x)};
		v[j] = v4i{(int)(threadIdx.x * 2246822519u + j), (int)(threadIdx.x * 3266489917u + j), (int)(blockIdx.x * 3 + j), (int)(j * 91 + threadIdx.x)};
	}
	const uint32_t msk = 0x11111111u << (threadIdx.x & 1);
	uint32_t x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
	for (int it = 0; it < iters; it++) {
		u[it & 3].x += it;
		v[it & 3].y ^= it;
		v4i B = and4(v[0], msk);
#pragma unroll
		for (int i = 0; i < 9; i++) {
			// the operand of MFMA i + 1 is formed (into another register set) while MFMA i runs
			v4i Bn = B;
			if (EXTRA >= 0) Bn = and4(v[(i + 1) & 3], msk);
#pragma unroll
			for (int e = 0; e < EXTRA; e++)
				x[e & 7] = __builtin_amdgcn_bitop3_b32(x[e & 7], (uint32_t)u[e & 3].x, msk, 0x96);
			__builtin_amdgcn_sched_barrier(0);
			if (CHAIN) MFMA4(acc[0], u[i & 3], B);
			else MFMA4(acc[i], u[i & 3], B);
			__builtin_amdgcn_sched_barrier(0);
			B = Bn;
		}
	}
	uint32_t s = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7];
#pragma unroll
	for (int i = 0; i < 9; i++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			s += (uint32_t)(int)acc[i][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAIN, int EXTRA, int WAVES>
static void run(const char *name, int blocks_per_cu)
{
	uint32_t *d;
	(void)hipMalloc(&d, 256 * 4 * 256 * 4);
	const int iters = 4000;
	hipEvent_t a, b;
	(void)hipEventCreate(&a);
	(void)hipEventCreate(&b);
	hipLaunchKernelGGL((k<CHAIN, EXTRA, WAVES>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, 10);
	(void)hipEventRecord(a);
	hipLaunchKernelGGL((k<CHAIN, EXTRA, WAVES>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, iters);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms;
	(void)hipEventElapsedTime(&ms, a, b);
	const double n_mfma_per_simd = (double)iters * 9 * blocks_per_cu;
	printf("%-46s %d waves/SIMD: %.3f ms, %.1f ns per MFMA per SIMD\n", name, blocks_per_cu, ms, ms * 1e6 / n_mfma_per_simd);
	(void)hipFree(d);
}

int main()
{
	run<0, -1, 2>("FP4, nine tiles, no VALU", 1);
	run<0, -1, 2>("FP4, nine tiles, no VALU", 2);
	run<0, -1, 3>("FP4, nine tiles, no VALU", 3);
	run<1, -1, 2>("FP4, one chain, no VALU", 1);
	run<1, -1, 2>("FP4, one chain, no VALU", 2);
	run<1, -1, 3>("FP4, one chain, no VALU", 3);
	run<0, 0, 2>("FP4, nine tiles, 4 VALU per MFMA", 2);
	run<0, 2, 2>("FP4, nine tiles, 6 VALU per MFMA", 2);
	run<0, 4, 2>("FP4, nine tiles, 8 VALU per MFMA", 2);
	run<0, 8, 2>("FP4, nine tiles, 12 VALU per MFMA", 2);
	run<0, 12, 2>("FP4, nine tiles, 16 VALU per MFMA", 2);
	run<1, 0, 2>("FP4, one chain, 4 VALU per MFMA", 2);
	run<1, 4, 2>("FP4, one chain, 8 VALU per MFMA", 2);
	run<1, 8, 2>("FP4, one chain, 12 VALU per MFMA", 2);
	run<0, 4, 3>("FP4, nine tiles, 8 VALU per MFMA", 3);
	run<1, 4, 3>("FP4, one chain, 8 VALU per MFMA", 3);
	run<0, 4, 2>("FP4, nine tiles, 8 VALU per MFMA", 1);
	return 0;
}
