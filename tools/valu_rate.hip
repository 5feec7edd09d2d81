// Microbenchmark: issue rate of 32-bit bitwise VALU ops on gfx950, forced with inline asm so the
// compiler cannot fold them.  8 independent chains per lane, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP2(name, str)                                                                         \
	struct name {                                                                              \
		static __device__ __forceinline__ void f(unsigned &a, unsigned b, unsigned c)          \
		{                                                                                      \
			asm volatile(str : "+v"(a) : "v"(b), "v"(c));                                      \
		}                                                                                      \
	};
OP2(op_xor, "v_xor_b32 %0, %0, %1")
OP2(op_and, "v_and_b32 %0, %0, %1")
OP2(op_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c")
OP2(op_perm, "v_perm_b32 %0, %0, %1, %2")
OP2(op_bfi, "v_bfi_b32 %0, %2, %0, %1")
OP2(op_and_or, "v_and_or_b32 %0, %0, %1, %2")
OP2(op_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
OP2(op_lshlrev, "v_lshlrev_b32 %0, 1, %0")
OP2(op_xor_e64, "v_xor_b32_e64 %0, %0, %1")
OP2(op_add, "v_add_u32 %0, %0, %1")
OP2(op_fma, "v_fma_f32 %0, %0, %1, %2")
OP2(op_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2") // placeholder: needs 64-bit regs, not run
template <class OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters)
{
	unsigned a[8];
	for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 17 + i;
	unsigned b = seed ^ 0x9e3779b9u + threadIdx.x, c = seed * 7u + 1;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 16; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) OP::f(a[i], b, c);
		}
	}
	unsigned s = 0;
	for (int i = 0; i < 8; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class OP>
void run(const char *name, int waves_per_simd = 8)
{
	unsigned *d;
	int blocks = 256 * waves_per_simd, iters = 2000;
	(void)hipMalloc(&d, blocks * 256 * 4);
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 10);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	double wave_instrs = (double)blocks * 4 * iters * 16 * 8;
	double per_simd_per_s = wave_instrs / (ms * 1e-3) / (256.0 * 4);
	printf("%d w/SIMD %-12s %8.3f ms  %.3e wave-instr/s/SIMD  (%.2f cycles/instr at 2.4 GHz)  %.1f T lane-ops/s\n", waves_per_simd, name, ms, per_simd_per_s,
	       2.4e9 / per_simd_per_s, wave_instrs * 64 / (ms * 1e-3) / 1e12);
	(void)hipFree(d);
}
int main()
{
	run<op_xor>("v_xor_b32");
	run<op_xor_e64>("v_xor_e64");
	run<op_and>("v_and_b32");
	run<op_add>("v_add_u32");
	run<op_lshlrev>("v_lshlrev");
	run<op_bitop3>("v_bitop3");
	run<op_perm>("v_perm");
	run<op_bfi>("v_bfi");
	run<op_and_or>("v_and_or");
	run<op_lshl_or>("v_lshl_or");
	run<op_fma>("v_fma_f32");
	for (int w = 1; w <= 8; w *= 2) {
		run<op_xor>("v_xor_b32", w);
		run<op_bitop3>("v_bitop3", w);
		run<op_perm>("v_perm", w);
	}
	return 0;
}
