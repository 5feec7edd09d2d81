// Experiment harness: variants of the fold (extrapolate_line) kernel at HBM-resident sizes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ibinius_amd/csrc tools/fold_variants.hip -o gpurun_out/fold_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ctable.hpp"
using namespace bn;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 nt_load(const uint4 *p)
{
	v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
	return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void nt_store(uint4 r, uint4 *p)
{
	v4u v = {r.x, r.y, r.z, r.w};
	__builtin_nontemporal_store(v, reinterpret_cast<v4u *>(p));
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_fold(uint4 *__restrict__ x0, const uint4 *__restrict__ x1, uint64_t n, f128 z)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	const uint64_t stride = (uint64_t)gridDim.x * 256;
	uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	for (; i + (U - 1) * stride < n; i += U * stride) {
		uint4 a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			if (NT) {
				a[u] = nt_load(&x0[i + u * stride]);
				b[u] = nt_load(&x1[i + u * stride]);
			} else {
				a[u] = x0[i + u * stride];
				b[u] = x1[i + u * stride];
			}
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			uint4 r = xor4(a[u], ctable_mul(tab, xor4(a[u], b[u])));
			if (NT) nt_store(r, &x0[i + u * stride]);
			else x0[i + u * stride] = r;
		}
	}
	for (; i < n; i += stride) {
		uint4 a = x0[i], b = x1[i];
		x0[i] = xor4(a, ctable_mul(tab, xor4(a, b)));
	}
}
// contiguous-chunk variant: each block owns a contiguous range (better DRAM page locality)
template <int U>
__global__ __launch_bounds__(256) void k_fold_chunk(uint4 *__restrict__ x0, const uint4 *__restrict__ x1, uint64_t n, f128 z)
{
	__shared__ ctable_smem tab;
	ctable_build(tab, z);
	const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
	const uint64_t lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
	for (uint64_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
		uint4 a[U], b[U];
#pragma unroll
		for (int u = 0; u < U; u++) {
			uint64_t k = i + u * 256;
			if (k < hi) { a[u] = x0[k]; b[u] = x1[k]; }
		}
#pragma unroll
		for (int u = 0; u < U; u++) {
			uint64_t k = i + u * 256;
			if (k < hi) x0[k] = xor4(a[u], ctable_mul(tab, xor4(a[u], b[u])));
		}
	}
}
__global__ void k_copy(uint4 *__restrict__ d, const uint4 *__restrict__ s, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}
__global__ void k_triad(uint4 *__restrict__ x0, const uint4 *__restrict__ x1, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) x0[i] = xor4(x0[i], x1[i]);
}
template <class F>
void timeit(const char *name, F launch, double bytes)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	launch();
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	for (int r = 0; r < 5; r++) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	ms /= 5;
	printf("%-28s %8.3f ms  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
}
int main()
{
	const uint64_t n = 1ull << 27; // output elements; 2 GiB + 2 GiB
	uint4 *x0, *x1;
	(void)hipMalloc(&x0, n * 16); (void)hipMalloc(&x1, n * 16);
	(void)hipMemset(x0, 0x5a, n * 16); (void)hipMemset(x1, 0xa5, n * 16);
	f128 z{0x123456789abcdef0ull, 0x0fedcba987654321ull};
	const double fold_bytes = 48.0 * n;
	timeit("copy (x0 <- x1)", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, x0, x1, n); }, 32.0 * n);
	timeit("xor triad (x0 ^= x1)", [&] { hipLaunchKernelGGL(k_triad, dim3(2048), dim3(256), 0, 0, x0, x1, n); }, 48.0 * n);
	for (unsigned g : {256u, 384u, 512u, 640u, 768u}) {
		char nm[64];
		snprintf(nm, 64, "fold U2 g=%u", g); timeit(nm, [&] { hipLaunchKernelGGL((k_fold<2, false>), dim3(g), dim3(256), 0, 0, x0, x1, n, z); }, fold_bytes);
		snprintf(nm, 64, "fold U2 NT g=%u", g); timeit(nm, [&] { hipLaunchKernelGGL((k_fold<2, true>), dim3(g), dim3(256), 0, 0, x0, x1, n, z); }, fold_bytes);
		snprintf(nm, 64, "fold U4 NT g=%u", g); timeit(nm, [&] { hipLaunchKernelGGL((k_fold<4, true>), dim3(g), dim3(256), 0, 0, x0, x1, n, z); }, fold_bytes);
		snprintf(nm, 64, "fold U8 NT g=%u", g); timeit(nm, [&] { hipLaunchKernelGGL((k_fold<8, true>), dim3(g), dim3(256), 0, 0, x0, x1, n, z); }, fold_bytes);
		snprintf(nm, 64, "fold U3 NT g=%u", g); timeit(nm, [&] { hipLaunchKernelGGL((k_fold<3, true>), dim3(g), dim3(256), 0, 0, x0, x1, n, z); }, fold_bytes);
	}
	for (unsigned g : {256u, 512u, 1024u, 2048u, 4096u}) {
		char nm[64];
		snprintf(nm, 64, "copy g=%u", g); timeit(nm, [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, x0, x1, n); }, 32.0 * n);
	}
	return 0;
}
