// tools/fp4_probe.hip -- operand / result layout, value decoding and issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 with
// both operands in FP4 (E2M1), probed on the device (no documentation in the image names the lane layout).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/fp4_probe.hip -o tools/fp4_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k_one(const uint32_t *a, const uint32_t *b, float *c)
{
	const unsigned l = threadIdx.x;
	v8i x = {(int)a[4 * l], (int)a[4 * l + 1], (int)a[4 * l + 2], (int)a[4 * l + 3], 0, 0, 0, 0};
	v8i y = {(int)b[4 * l], (int)b[4 * l + 1], (int)b[4 * l + 2], (int)b[4 * l + 3], 0, 0, 0, 0};
	v16f acc = {0};
	acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
	for (int r = 0; r < 16; r++) c[16 * l + r] = acc[r];
}

template <int FP4>
__global__ void k_rate(float *out, int iters)
{
	v8i x = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, y = {5, 6, 7, (int)threadIdx.x, 0, 0, 0, 0};
	v16f acc[4] = {};
	v16i iacc[4] = {};
	for (int i = 0; i < iters; i++) {
#pragma unroll
		for (int t = 0; t < 4; t++) {
			if (FP4)
				acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, acc[t], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
			else
				iacc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(v4i{x[0], x[1], x[2], x[3]}, v4i{y[0], y[1], y[2], y[3]}, iacc[t], 0, 0, 0);
		}
	}
	float s = 0;
	for (int t = 0; t < 4; t++)
		for (int r = 0; r < 16; r++) s += acc[t][r] + (float)iacc[t][r];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
	uint32_t *d_a, *d_b;
	float *d_c;
	(void)hipMalloc(&d_a, 64 * 16);
	(void)hipMalloc(&d_b, 64 * 16);
	(void)hipMalloc(&d_c, 64 * 16 * 4);
	std::vector<uint32_t> ha(256), hb(256);
	std::vector<float> hc(1024);
	auto run = [&]() {
		(void)hipMemcpy(d_a, ha.data(), 1024, hipMemcpyHostToDevice);
		(void)hipMemcpy(d_b, hb.data(), 1024, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, d_a, d_b, d_c);
		(void)hipMemcpy(hc.data(), d_c, 4096, hipMemcpyDeviceToHost);
	};
	// 1. value decoding: A = code in every nibble of row 0's k = 0 entry only; B = 1.0 (0x2) everywhere
	printf("value decoding (A nibble code -> product with 1.0):");
	for (unsigned code = 0; code < 16; code++) {
		std::fill(ha.begin(), ha.end(), 0u);
		std::fill(hb.begin(), hb.end(), 0x22222222u);
		ha[0] = code; // lane 0, register 0, nibble 0
		run();
		float v = 0;
		for (int i = 0; i < 1024; i++)
			if (hc[i] != 0) v = hc[i];
		printf(" %x:%g", code, v);
	}
	printf("\n");
	// 2. A layout: one nibble = 1.0 at (lane, reg, nib); B all ones -> which C row lights up (all columns); and K position
	//    via a B that is 1.0 only at one k for all columns
	auto crow = [&](int *row_out) {
		// C/D layout of the 32x32 f32 result: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
		int row = -1, cnt = 0;
		for (int l = 0; l < 64; l++)
			for (int r = 0; r < 16; r++)
				if (hc[16 * l + r] != 0) {
					cnt++;
					row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
				}
		*row_out = row;
		return cnt;
	};
	printf("A operand: (lane, reg, nibble) -> row, with B = all ones:\n");
	for (int l : {0, 1, 31, 32, 33, 63})
		for (int rg : {0, 3})
			for (int nb : {0, 7}) {
				std::fill(ha.begin(), ha.end(), 0u);
				std::fill(hb.begin(), hb.end(), 0x22222222u);
				ha[4 * l + rg] = 0x2u << (4 * nb);
				run();
				int row, cnt = crow(&row);
				printf("  lane %2d reg %d nib %d -> row %2d (%d nonzero entries)\n", l, rg, nb, row, cnt);
			}
	// 3. K index of (lane, reg, nibble): A(lane la, reg ra, nib na) = 1 and B(lane lb, reg rb, nib nb) = 1 give a nonzero
	//    product iff they share k.  For A at lane 0 / lane 32, scan all B positions of column 0 (lanes 0 and 32).
	printf("K pairing: A position -> the B position (same k) in column 0:\n");
	for (int la : {0, 32})
		for (int ra : {0, 1, 3})
			for (int na : {0, 1, 7}) {
				std::fill(ha.begin(), ha.end(), 0u);
				ha[4 * la + ra] = 0x2u << (4 * na);
				for (int lb : {0, 32})
					for (int rb = 0; rb < 4; rb++)
						for (int nb = 0; nb < 8; nb++) {
							std::fill(hb.begin(), hb.end(), 0u);
							hb[4 * lb + rb] = 0x2u << (4 * nb);
							run();
							bool any = false;
							for (int i = 0; i < 1024; i++)
								if (hc[i] != 0) any = true;
							if (any) printf("  A(lane %2d reg %d nib %d) <-> B(lane %2d reg %d nib %d)\n", la, ra, na, lb, rb, nb);
						}
			}
	// 4. exact accumulation of many small terms: all A = 0.5 (0x1), all B = 0.5: every entry = 64 * 0.25 = 16
	std::fill(ha.begin(), ha.end(), 0x11111111u);
	std::fill(hb.begin(), hb.end(), 0x11111111u);
	run();
	printf("all 0.5 x 0.5 over K = 64: C[0] = %g (expect 16)\n", hc[0]);
	// 5. issue rate
	float *d_o;
	(void)hipMalloc(&d_o, 1024 * 256 * 4);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	for (int fp4 = 0; fp4 < 2; fp4++) {
		const int iters = 20000;
		for (int rep = 0; rep < 2; rep++) {
			(void)hipEventRecord(e0, 0);
			if (fp4)
				hipLaunchKernelGGL(k_rate<1>, dim3(1024), dim3(256), 0, 0, d_o, iters);
			else
				hipLaunchKernelGGL(k_rate<0>, dim3(1024), dim3(256), 0, 0, d_o, iters);
			(void)hipEventRecord(e1, 0);
			(void)hipEventSynchronize(e1);
			float ms;
			(void)hipEventElapsedTime(&ms, e0, e1);
			// 1024 WGs x 4 waves over 1024 SIMDs: 4 waves per SIMD, each iters * 4 MFMAs
			const double per_mfma_ns = ms * 1e6 / ((double)iters * 4 * 4);
			if (rep) printf("%s: %.2f ns per MFMA per SIMD (%.0f T MAC/s chip-wide)\n", fp4 ? "fp4 32x32x64" : "i8  32x32x32", per_mfma_ns,
			                (fp4 ? 65536.0 : 32768.0) / per_mfma_ns * 1024 / 1e3);
		}
	}
	return 0;
}
