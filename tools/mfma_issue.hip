// tools/mfma_issue.hip -- how do v_mfma_i32_32x32x32_i8 and the VALU that prepares its operands share a
// SIMD?  Nine independent accumulator tiles per iteration; K bitwise VALU per operand register set.
//   PIPE 0: operands of MFMA i are produced right before MFMA i (same destination registers every time)
//   PIPE 1: operands of MFMA i+1 are produced (into a second register set) before MFMA i is issued
//   PIPE 2: no VALU at all (pure MFMA rate)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_issue.hip -o tools/mfma_issue
#include <hip/hip_runtime.h>

#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v4i and4(v4i x, uint32_t m)
{
	return v4i{(int)((uint32_t)x.x & m), (int)((uint32_t)x.y & m), (int)((uint32_t)x.z & m), (int)((uint32_t)x.w & m)};
}

template <int PIPE, int WAVES, int EXTRA = 0>
__global__ __launch_bounds__(256, WAVES) void k(uint32_t *out, int iters)
{
	v16i acc[9];
#pragma unroll
	for (int i = 0; i < 9; i++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[i][r] = 0;
	v4i u[4], v[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		u[j] = v4i{(int)(threadIdx.x * 2654435761u + j), (int)(threadIdx.x * 40503u + j), (int)(blockIdx.x + j), (int)(j * 77 + threadIdx.x)};
		v[j] = v4i{(int)(threadIdx.x * 2246822519u + j), (int)(threadIdx.x * 3266489917u + j), (int)(blockIdx.x * 3 + j), (int)(j * 91 + threadIdx.x)};
	}
	const uint32_t msk = 0x01010101u << (threadIdx.x & 7);
	uint32_t x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
	for (int it = 0; it < iters; it++) {
		// stand-in for the LDS reads: the raw words change every iteration
		u[it & 3].x += it;
		v[it & 3].y ^= it;
		if (PIPE == 2) {
#pragma unroll
			for (int i = 0; i < 9; i++)
				acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(u[i & 3], v[(i + 1) & 3], acc[i], 0, 0, 0);
		} else if (PIPE == 0) {
#pragma unroll
			for (int i = 0; i < 9; i++) {
				const v4i A = and4(u[i & 3] ^ u[(i >> 2) & 3], msk), B = and4(v[i & 3] ^ v[(i >> 2) & 3], msk);
#pragma unroll
				for (int e = 0; e < EXTRA; e++) // independent bitwise VALU riding along
					x[e & 7] = __builtin_amdgcn_bitop3_b32(x[e & 7], (uint32_t)u[e & 3].x, msk, 0x96);
				__builtin_amdgcn_sched_barrier(0);
				acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[i], 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
			}
		} else {
			v4i A = and4(u[0], msk), B = and4(v[0], msk);
#pragma unroll
			for (int i = 0; i < 9; i++) {
				v4i An = A, Bn = B;
				if (i < 8) {
					An = and4(u[(i + 1) & 3] ^ u[((i + 1) >> 2) & 3], msk);
					Bn = and4(v[(i + 1) & 3] ^ v[((i + 1) >> 2) & 3], msk);
				}
				__builtin_amdgcn_sched_barrier(0);
				acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[i], 0, 0, 0);
				__builtin_amdgcn_sched_barrier(0);
				A = An;
				B = Bn;
			}
		}
	}
	uint32_t s = x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7];
#pragma unroll
	for (int i = 0; i < 9; i++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			s ^= (uint32_t)acc[i][r];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PIPE, int WAVES, int EXTRA = 0>
static void run(const char *name, int blocks_per_cu)
{
	uint32_t *d;
	(void)hipMalloc(&d, 256 * 4 * 256 * 4);
	const int iters = 4000;
	hipEvent_t a, b;
	(void)hipEventCreate(&a);
	(void)hipEventCreate(&b);
	hipLaunchKernelGGL((k<PIPE, WAVES, EXTRA>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, 10);
	(void)hipEventRecord(a);
	hipLaunchKernelGGL((k<PIPE, WAVES, EXTRA>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, iters);
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms;
	(void)hipEventElapsedTime(&ms, a, b);
	const double n_mfma_per_simd = (double)iters * 9 * blocks_per_cu;
	printf("%-34s %d waves/SIMD: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TOPS\n", name, blocks_per_cu, ms, ms * 1e6 / n_mfma_per_simd,
	       n_mfma_per_simd * 1024 * 65536.0 / (ms * 1e-3) * 1e-12);
	(void)hipFree(d);
}

int main()
{
	run<2, 2>("pure MFMA", 1);
	run<2, 2>("pure MFMA", 2);
	run<0, 2>("operands right before each MFMA", 1);
	run<0, 2>("operands right before each MFMA", 2);
	run<1, 2>("operands one MFMA ahead", 1);
	run<1, 2>("operands one MFMA ahead", 2);
	run<0, 2, 2>("8 + 2 VALU per MFMA", 2);
	run<0, 2, 4>("8 + 4 VALU per MFMA", 2);
	run<0, 2, 6>("8 + 6 VALU per MFMA", 2);
	run<0, 2, 8>("8 + 8 VALU per MFMA", 2);
	run<0, 2, 12>("8 + 12 VALU per MFMA", 2);
	run<0, 2, 16>("8 + 16 VALU per MFMA", 2);
	run<0, 2, 4>("8 + 4 VALU per MFMA", 1);
	run<0, 2, 8>("8 + 8 VALU per MFMA", 1);
	return 0;
}
