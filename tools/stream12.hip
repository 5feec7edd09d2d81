// Experiment harness (round 4): what the HBM system gives the access pattern of the fused fold + evaluate kernel
// (kernels_foldeval_mfma.hip) when NOTHING is computed: per array, reads x[p], x[p + n], x[p + 2n], x[p + 3n] (n = N/4) and writes
// f[p], f[p + n] -- eight read streams and four write streams over the two arrays -- against the fold-alone pattern (four read
// + two write streams), a plain copy and an XOR triad.  Variants: tile order (strided / XCD-aware / contiguous chunk per
// workgroup), workgroups per CU, tiles in flight per lane, points per workgroup iteration, non-temporal accesses, in place or not.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream12.hip -o tools/stream12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ v4u ld(const v4u *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v4u v, v4u *p)
{
	if (NT) __builtin_nontemporal_store(v, p);
	else *p = v;
}

struct sargs {
	const v4u *x[2]; // the two arrays, N elements each
	v4u *out[2];     // N/2 elements each (may be x)
	uint64_t n;      // N/4
	int order;       // 0 strided, 1 XCD-aware, 2 contiguous chunk per workgroup, 3 XCD-aware + contiguous inside the XCD's eighth
};

// W = consecutive 256-point tiles a workgroup takes per iteration (a lane handles W points, 256 apart), D = 1: loads of the next
// iteration issued before the stores of the current one (register double buffer)
template <int W, bool NT, int WGS>
__global__ __launch_bounds__(256, WGS) void k_s12(sargs a)
{
	const uint64_t n = a.n, n_tiles = n / (256 * W);
	uint64_t tbase = 0, tstride = gridDim.x, tlimit = n_tiles, t0 = blockIdx.x;
	if (a.order == 1 || a.order == 3) {
		const uint64_t chunk = (n_tiles + 7) >> 3;
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		tlimit = tbase >= n_tiles ? 0 : (n_tiles - tbase < chunk ? n_tiles - tbase : chunk);
		if (a.order == 3) {
			const uint64_t per = (tlimit + tstride - 1) / tstride;
			tbase += t0 * per;
			tlimit = t0 * per >= tlimit ? 0 : (tlimit - t0 * per < per ? tlimit - t0 * per : per);
			t0 = 0;
			tstride = 1;
		}
	} else if (a.order == 2) {
		const uint64_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
		tbase = blockIdx.x * per;
		tlimit = tbase >= n_tiles ? 0 : (n_tiles - tbase < per ? n_tiles - tbase : per);
		t0 = 0;
		tstride = 1;
	}
	v4u c[W][8], nx[W][8];
	auto load = [&](uint64_t t, v4u(&r)[W][8]) {
#pragma unroll
		for (int w = 0; w < W; w++) {
			const uint64_t p = ((tbase + t) * W + w) * 256 + threadIdx.x;
#pragma unroll
			for (int k = 0; k < 4; k++) { // quadrant = 2 * array + half
				const uint64_t e = (k & 1 ? n : 0) + p;
				r[w][2 * k] = ld<NT>(&a.x[k >> 1][e]);
				r[w][2 * k + 1] = ld<NT>(&a.x[k >> 1][e + 2 * n]);
			}
		}
	};
	uint64_t t = t0;
	if (t >= tlimit) return;
	load(t, c);
	for (;;) {
		const uint64_t tn = t + tstride;
		const bool more = tn < tlimit;
		if (more) load(tn, nx);
#pragma unroll
		for (int w = 0; w < W; w++) {
			const uint64_t p = ((tbase + t) * W + w) * 256 + threadIdx.x;
#pragma unroll
			for (int k = 0; k < 4; k++)
				st<NT>(c[w][2 * k] ^ c[w][2 * k + 1], &a.out[k >> 1][(k & 1 ? n : 0) + p]);
		}
		if (!more) break;
#pragma unroll
		for (int w = 0; w < W; w++)
#pragma unroll
			for (int k = 0; k < 8; k++)
				c[w][k] = nx[w][k];
		t = tn;
	}
}

// the fold-alone pattern over both arrays: out[i] = x[i] ^ x[i + N/2], i < N/2 (four read + two write streams)
template <int W, bool NT, int WGS>
__global__ __launch_bounds__(256, WGS) void k_s6(sargs a)
{
	const uint64_t h = 2 * a.n, n_tiles = h / (256 * W);
	uint64_t tbase = 0, tstride = gridDim.x, tlimit = n_tiles, t0 = blockIdx.x;
	if (a.order == 1) {
		const uint64_t chunk = (n_tiles + 7) >> 3;
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		tlimit = tbase >= n_tiles ? 0 : (n_tiles - tbase < chunk ? n_tiles - tbase : chunk);
	}
	for (uint64_t t = t0; t < tlimit; t += tstride) {
		v4u c[W][4];
#pragma unroll
		for (int w = 0; w < W; w++) {
			const uint64_t p = ((tbase + t) * W + w) * 256 + threadIdx.x;
#pragma unroll
			for (int k = 0; k < 2; k++) {
				c[w][2 * k] = ld<NT>(&a.x[k][p]);
				c[w][2 * k + 1] = ld<NT>(&a.x[k][p + h]);
			}
		}
#pragma unroll
		for (int w = 0; w < W; w++) {
			const uint64_t p = ((tbase + t) * W + w) * 256 + threadIdx.x;
#pragma unroll
			for (int k = 0; k < 2; k++)
				st<NT>(c[w][2 * k] ^ c[w][2 * k + 1], &a.out[k][p]);
		}
	}
}
// the round-0 pattern: every element of both arrays read once, as four streams x[k][p], x[k][p + N/2] (no stores)
template <int W, bool NT, int WGS>
__global__ __launch_bounds__(256, WGS) void k_r4(sargs a)
{
	const uint64_t h = 2 * a.n, n_tiles = h / (256 * W);
	uint64_t tbase = 0, tstride = gridDim.x, tlimit = n_tiles, t0 = blockIdx.x;
	if (a.order == 1) {
		const uint64_t chunk = (n_tiles + 7) >> 3;
		tbase = (blockIdx.x & 7) * chunk;
		tstride = gridDim.x >> 3;
		t0 = blockIdx.x >> 3;
		tlimit = tbase >= n_tiles ? 0 : (n_tiles - tbase < chunk ? n_tiles - tbase : chunk);
	}
	v4u acc = {0, 0, 0, 0};
	for (uint64_t t = t0; t < tlimit; t += tstride) {
#pragma unroll
		for (int w = 0; w < W; w++) {
			const uint64_t p = ((tbase + t) * W + w) * 256 + threadIdx.x;
#pragma unroll
			for (int k = 0; k < 2; k++)
				acc ^= ld<NT>(&a.x[k][p]) ^ ld<NT>(&a.x[k][p + h]);
		}
	}
	if (acc.x == 0x12345 && acc.y == 0x777) a.out[0][0] = acc;
}
__global__ void k_copy(v4u *__restrict__ d, const v4u *__restrict__ s, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}
__global__ void k_read(v4u *__restrict__ d, const v4u *__restrict__ s, uint64_t n)
{
	v4u acc = {0, 0, 0, 0};
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(&s[i]);
	if (acc.x == 0x12345 && acc.y == 0x777) d[0] = acc;
}

template <class F>
static double timeit(const char *name, F launch, double bytes, int reps = 6)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	launch();
	launch();
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	for (int r = 0; r < reps; r++) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	ms /= reps;
	printf("%-58s %8.3f ms  %7.1f GB/s  %.3f\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
	fflush(stdout);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return ms;
}

int main(int argc, char **argv)
{
	const int lg = argc > 1 ? atoi(argv[1]) : 27;
	const uint64_t N = 1ull << lg; // elements per array
	int n_cu = 256;
	hipDeviceProp_t pr;
	if (hipGetDeviceProperties(&pr, 0) == hipSuccess) n_cu = pr.multiProcessorCount;
	v4u *x[2], *o[2];
	for (int k = 0; k < 2; k++) {
		if (hipMalloc(&x[k], N * 16) != hipSuccess || hipMalloc(&o[k], N * 8) != hipSuccess) return 1;
		(void)hipMemset(x[k], 0x5a + k, N * 16);
	}
	printf("N = 2^%d elements per array, %d CUs\n", lg, n_cu);
	const double B = 48.0 * N; // 24 * m * N
	timeit("copy 2 GiB-class (x1 <- x0), 2048 wgs", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, x[1], x[0], N); }, 32.0 * N);
	timeit("read only, 2048 wgs", [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, o[0], x[0], N); }, 16.0 * N);
	if (argc > 2) { // round-0 pattern only
		sargs a{};
		a.x[0] = x[0];
		a.x[1] = x[1];
		a.out[0] = o[0];
		a.out[1] = o[1];
		a.n = N / 4;
		char nm[128];
#define RUNR4(W, NT, WGS, ORD)                                                                                                 \
	a.order = ORD;                                                                                                             \
	snprintf(nm, 128, "r4  W=%d NT=%d wgs/CU=%d order=%d", W, NT, WGS, ORD);                                                  \
	timeit(nm, [&] { hipLaunchKernelGGL((k_r4<W, NT, WGS>), dim3(n_cu * WGS), dim3(256), 0, 0, a); }, 32.0 * N);
		RUNR4(1, false, 2, 0) RUNR4(1, true, 2, 0) RUNR4(1, true, 2, 1) RUNR4(1, true, 4, 1) RUNR4(1, true, 8, 1) RUNR4(2, true, 4, 1) RUNR4(2, true, 8, 1)
		RUNR4(4, true, 4, 1) RUNR4(4, true, 2, 1) RUNR4(4, true, 1, 1) RUNR4(2, true, 4, 0) RUNR4(1, false, 8, 1)
		return 0;
	}
	for (int inplace = 0; inplace < 2; inplace++) {
		sargs a{};
		a.x[0] = x[0];
		a.x[1] = x[1];
		a.out[0] = inplace ? x[0] : o[0];
		a.out[1] = inplace ? x[1] : o[1];
		a.n = N / 4;
		char nm[128];
#define RUN6(W, NT, WGS, ORD)                                                                                                  \
	a.order = ORD;                                                                                                             \
	snprintf(nm, 128, "s6  %s W=%d NT=%d wgs/CU=%d order=%d", inplace ? "inplace" : "outofpl", W, NT, WGS, ORD);              \
	timeit(nm, [&] { hipLaunchKernelGGL((k_s6<W, NT, WGS>), dim3(n_cu * WGS), dim3(256), 0, 0, a); }, B);
#define RUN12(W, NT, WGS, ORD)                                                                                                 \
	a.order = ORD;                                                                                                             \
	snprintf(nm, 128, "s12 %s W=%d NT=%d wgs/CU=%d order=%d", inplace ? "inplace" : "outofpl", W, NT, WGS, ORD);              \
	timeit(nm, [&] { hipLaunchKernelGGL((k_s12<W, NT, WGS>), dim3(n_cu * WGS), dim3(256), 0, 0, a); }, B);
		RUN6(1, false, 2, 0) RUN6(1, true, 2, 0) RUN6(2, true, 2, 0) RUN6(2, true, 2, 1) RUN6(2, true, 4, 0) RUN6(4, true, 2, 0)
		RUN12(1, false, 2, 0) RUN12(1, false, 2, 1) RUN12(1, true, 2, 0) RUN12(1, true, 2, 1) RUN12(1, true, 2, 2) RUN12(1, true, 2, 3)
		RUN12(1, false, 4, 1) RUN12(1, true, 4, 1) RUN12(1, true, 4, 0) RUN12(1, true, 8, 1)
		RUN12(2, true, 2, 1) RUN12(2, true, 2, 0) RUN12(2, true, 4, 1) RUN12(4, true, 2, 1) RUN12(4, true, 1, 1) RUN12(2, true, 1, 1) RUN12(1, true, 1, 1)
	}
	return 0;
}
