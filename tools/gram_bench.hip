// tools/gram_bench.hip -- stand-alone check + timing of the matrix-core round-evaluation kernels
// (kernels_roundeval_mfma.hip) against a host bilinear-walk product.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude tools/gram_bench.hip -o tools/gram_bench
#include "../binius_amd/csrc/kernels_roundeval_mfma.hip"
#include "../binius_amd/csrc/kernels_foldeval_mfma.hip"
#include "../binius_amd/csrc/kernels_roundeval_fp4.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bn;

// every element different (SplitMix64 of its index)
__global__ void k_fill_random(f128 *p, uint64_t n, uint64_t seed)
{
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
		uint64_t z = seed + 2 * i * 0x9E3779B97F4A7C15ull, w[2];
		for (int k = 0; k < 2; k++) {
			z += 0x9E3779B97F4A7C15ull;
			uint64_t x = z;
			x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
			x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
			w[k] = x ^ (x >> 31);
		}
		p[i] = f128{w[0], w[1]};
	}
}

static uint64_t sm64(uint64_t &s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
	setenv("BN_FP4", "0", 1); // launch_roundeval_mfma_pair stays the int8 kernel here: it is the reference of the FP4 one
	const int log_n = argc > 1 ? atoi(argv[1]) : 24;
	const uint64_t n = 1ull << log_n;
	std::vector<f128> h[4];
	uint64_t seed = 1234;
	f128 *d[4];
	const uint64_t n_host = n < (1ull << 18) ? n : (1ull << 18);
	for (int k = 0; k < 4; k++) {
		h[k].resize(n_host);
		for (uint64_t i = 0; i < n_host; i++) h[k][i] = f128{sm64(seed), sm64(seed)};
		// GRAM_ONE_ARENA=1: the four arrays back to back in ONE allocation (the way a prover's arena lays them out)
		static f128 *arena = nullptr;
		if (getenv("GRAM_ONE_ARENA")) {
			if (!arena) (void)hipMalloc(&arena, 4 * n * 16);
			d[k] = arena + k * n;
		} else {
			(void)hipMalloc(&d[k], n * 16);
		}
		for (uint64_t off = 0; off < n; off += n_host) // repeat the host block
			(void)hipMemcpy(d[k] + off, h[k].data(), n_host * 16, hipMemcpyHostToDevice);
	}
	// Beyond the host block the arrays are filled on the device with distinct values.  (Until late in round 2 the host block
	// was simply repeated: then x[i] == x[i + N/2] for every N/2 that is a multiple of the block, the fused kernel folds
	// x0 + z * 0, every table lookup of the constant multiplication hits entry 0 -- one broadcast -- and the kernel looks
	// 15 % faster than it is on a prover's data.  GRAM_REPEAT=1 brings that data back.)
	if (!getenv("GRAM_REPEAT") && n > n_host)
		for (int k = 0; k < 4; k++)
			hipLaunchKernelGGL(k_fill_random, dim3(4096), dim3(256), 0, 0, d[k] + n_host, n - n_host, 0x1234567ull * (k + 1));
	f128 *d_out;
	(void)hipMalloc(&d_out, 32);
	int bad = 0;
	const bool prof = argc > 2; // profile mode: one launch of each kernel at full size, nothing else
	for (uint64_t nc : {1ull, 5ull, 255ull, 256ull, 257ull, 1000ull, 4096ull, 70001ull, 1ull << 17}) {
		if (nc > n_host || prof) continue;
		(void)hipMemset(d_out, 0, 32);
		hipError_t e = launch_roundeval_mfma_pair(0, 256, d[0], d[1], d[2], d[3], nc, d_out, nullptr);
		f128 got[2];
		(void)hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost);
		f128 e1 = f128_zero(), ei = f128_zero();
		for (uint64_t i = 0; i < nc; i++) {
			e1 ^= mul_slow(h[0][i], h[2][i]);
			ei ^= mul_slow(h[0][i] ^ h[1][i], h[2][i] ^ h[3][i]);
		}
		const bool ok = got[0] == e1 && got[1] == ei && e == hipSuccess;
		bad += !ok;
		printf("pair  n=%llu: %s\n", (unsigned long long)nc, ok ? "OK" : "MISMATCH");
		{
			// the same sums on the FP4 matrix path
			(void)hipMemset(d_out, 0, 32);
			const hipError_t e4 = launch_roundeval_fp4_pair(0, 256, d[0], d[1], d[2], d[3], nc, d_out, nullptr);
			f128 g4[2];
			(void)hipMemcpy(g4, d_out, 32, hipMemcpyDeviceToHost);
			const bool ok4 = g4[0] == e1 && g4[1] == ei && e4 == hipSuccess;
			bad += !ok4;
			printf("fp4   n=%llu: %s\n", (unsigned long long)nc, ok4 ? "OK" : "MISMATCH");
		}
		// split: sums over [0,nc/2) and [nc/2, nc/2*2)
		const uint64_t hn = nc / 2;
		if (hn) {
			(void)hipMemset(d_out, 0, 32);
			e = launch_roundeval_mfma_split(0, 256, d[0], d[2], hn, hn, d_out);
			(void)hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost);
			f128 s0 = f128_zero(), s1 = f128_zero();
			for (uint64_t i = 0; i < hn; i++) {
				s0 ^= mul_slow(h[0][i], h[2][i]);
				s1 ^= mul_slow(h[0][i + hn], h[2][i + hn]);
			}
			const bool ok2 = got[0] == s0 && got[1] == s1 && e == hipSuccess;
			bad += !ok2;
			printf("split n=%llu: %s\n", (unsigned long long)hn, ok2 ? "OK" : "MISMATCH");
		}
	}
	hipEvent_t ea, eb;
	(void)hipEventCreate(&ea);
	(void)hipEventCreate(&eb);
	for (int rep = 0; rep < (prof ? 3 : 4); rep++) {
		(void)hipMemset(d_out, 0, 32);
		(void)hipEventRecord(ea);
		(void)launch_roundeval_mfma_pair(0, 256, d[0], d[1], d[2], d[3], n, d_out, nullptr);
		(void)hipEventRecord(eb);
		(void)hipEventSynchronize(eb);
		float ms;
		(void)hipEventElapsedTime(&ms, ea, eb);
		printf("pair n=2^%d: %.3f ms  %.2f G points/s  %.2f TB/s algorithmic (64 B/point)\n", log_n, ms, n / ms * 1e-6, n * 64.0 / ms * 1e-9);
	}
	{
		// FP4 form against the int8 form at full size (the host product is too slow there), then its timing
		f128 ref[2], g4[2];
		(void)hipMemset(d_out, 0, 32);
		(void)launch_roundeval_mfma_pair(0, 256, d[0], d[1], d[2], d[3], n, d_out, nullptr);
		(void)hipMemcpy(ref, d_out, 32, hipMemcpyDeviceToHost);
		(void)hipMemset(d_out, 0, 32);
		(void)launch_roundeval_fp4_pair(0, 256, d[0], d[1], d[2], d[3], n, d_out, nullptr);
		(void)hipMemcpy(g4, d_out, 32, hipMemcpyDeviceToHost);
		const bool ok = ref[0] == g4[0] && ref[1] == g4[1];
		bad += !ok;
		printf("fp4 vs int8 at n=2^%d: %s\n", log_n, ok ? "OK" : "MISMATCH");
#ifdef BN_FP4_PHASES
		{
			unsigned long long z8[8] = {0}, c8[8];
			(void)hipMemcpyToSymbol(HIP_SYMBOL(bn::fp4_phase_cycles), z8, sizeof(z8));
			(void)hipMemset(d_out, 0, 32);
			(void)launch_roundeval_fp4_pair(0, 256, d[0], d[1], d[2], d[3], n, d_out, nullptr);
			(void)hipDeviceSynchronize();
			(void)hipMemcpyFromSymbol(c8, HIP_SYMBOL(bn::fp4_phase_cycles), sizeof(c8));
			const double tiles = (double)((n + 255) / 256) / 512.0;
			printf("fp4 phases, cycles per tile (workgroup 0, wave 0): wait loads %.0f, barrier 1 %.0f, stage %.0f, barrier 2 %.0f, gram %.0f\n",
			       c8[0] / tiles, c8[1] / tiles, c8[2] / tiles, c8[3] / tiles, c8[4] / tiles);
		}
#endif
		for (int rep = 0; rep < (prof ? 3 : 4); rep++) {
			hipEvent_t ea, eb;
			(void)hipEventCreate(&ea);
			(void)hipEventCreate(&eb);
			(void)hipMemset(d_out, 0, 32);
			(void)hipEventRecord(ea);
			(void)launch_roundeval_fp4_pair(0, 256, d[0], d[1], d[2], d[3], n, d_out, nullptr);
			(void)hipEventRecord(eb);
			(void)hipEventSynchronize(eb);
			float ms;
			(void)hipEventElapsedTime(&ms, ea, eb);
			printf("fp4  n=2^%d: %.3f ms  %.2f G points/s  %.2f TB/s algorithmic (64 B/point)\n", log_n, ms, n / ms * 1e-6, n * 64.0 / ms * 1e-9);
		}
	}
	// ---- fused fold + evaluation: arrays a = d[0], b = d[2] of N elements, folded into d[1], d[3]
	const f128 z{0x0123456789abcdefull, 0xfedcba9876543210ull};
	for (uint64_t N : {1024ull, 4096ull + 8, 1ull << 16, 1ull << 18}) {
		if (N > n_host || prof) continue;
		foldeval_args fa{};
		fa.x0[0] = d[0]; fa.x1[0] = d[0] + N / 2; fa.out[0] = d[1];
		fa.x0[1] = d[2]; fa.x1[1] = d[2] + N / 2; fa.out[1] = d[3];
		(void)hipMemset(d_out, 0, 32);
		hipError_t e = launch_foldeval_mfma(0, 256, fa, N, z, d_out, nullptr);
		f128 got[2];
		(void)hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost);
		std::vector<f128> fa_h(N / 2), fb_h(N / 2), ga(N / 2), gb(N / 2);
		for (uint64_t i = 0; i < N / 2; i++) {
			fa_h[i] = h[0][i] ^ mul_slow(h[0][i] ^ h[0][i + N / 2], z);
			fb_h[i] = h[2][i] ^ mul_slow(h[2][i] ^ h[2][i + N / 2], z);
		}
		(void)hipMemcpy(ga.data(), d[1], N / 2 * 16, hipMemcpyDeviceToHost);
		(void)hipMemcpy(gb.data(), d[3], N / 2 * 16, hipMemcpyDeviceToHost);
		f128 e1 = f128_zero(), ei = f128_zero();
		const uint64_t q = N / 4;
		for (uint64_t j = 0; j < q; j++) {
			e1 ^= mul_slow(fa_h[j + q], fb_h[j + q]);
			ei ^= mul_slow(fa_h[j] ^ fa_h[j + q], fb_h[j] ^ fb_h[j + q]);
		}
		bool same = true;
		for (uint64_t i = 0; i < N / 2; i++) same = same && ga[i] == fa_h[i] && gb[i] == fb_h[i];
		const bool ok = same && got[0] == e1 && got[1] == ei && e == hipSuccess;
		bad += !ok;
		printf("fused N=%llu: fold %s, sums %s\n", (unsigned long long)N, same ? "OK" : "MISMATCH", (got[0] == e1 && got[1] == ei) ? "OK" : "MISMATCH");
	}
	{
		// timing: in place on a = d[0], b = d[2] (N = n elements each)
		foldeval_args fa{};
		fa.x0[0] = d[0]; fa.x1[0] = d[0] + n / 2; fa.out[0] = d[0];
		fa.x0[1] = d[2]; fa.x1[1] = d[2] + n / 2; fa.out[1] = d[2];
		for (int rep = 0; rep < (prof ? 3 : 4); rep++) {
			(void)hipMemset(d_out, 0, 32);
			(void)hipEventRecord(ea);
			(void)launch_foldeval_mfma(0, 256, fa, n, z, d_out, nullptr);
			(void)hipEventRecord(eb);
			(void)hipEventSynchronize(eb);
			float ms;
			(void)hipEventElapsedTime(&ms, ea, eb);
			printf("fused N=2^%d: %.3f ms  %.2f TB/s algorithmic (48 B/element)\n", log_n, ms, n * 48.0 / ms * 1e-9);
		}
		if (argc > 4) {
			// interleaved with the round-evaluation kernel, the way a step of the prover alternates them: does one kernel's
			// power draw cost the next one its clock?
			hipEvent_t e[4];
			for (auto &x : e) (void)hipEventCreate(&x);
			for (int it = 0; it < atoi(argv[4]); it++) {
				(void)hipEventRecord(e[0]);
				(void)launch_roundeval_fp4_pair(0, 256, d[0], d[1], d[2], d[3], n, d_out, nullptr);
				(void)hipEventRecord(e[1]);
				(void)launch_foldeval_mfma(0, 256, fa, n, z, d_out, nullptr);
				(void)hipEventRecord(e[2]);
				(void)launch_foldeval_mfma(0, 256, fa, n, z, d_out, nullptr);
				(void)hipEventRecord(e[3]);
				(void)hipEventSynchronize(e[3]);
				float m0, m1, m2;
				(void)hipEventElapsedTime(&m0, e[0], e[1]);
				(void)hipEventElapsedTime(&m1, e[1], e[2]);
				(void)hipEventElapsedTime(&m2, e[2], e[3]);
				printf("interleaved %d: fp4 round eval %.3f ms, fused %.3f ms, fused again %.3f ms\n", it, m0, m1, m2);
			}
		}
		if (argc > 3) {
			// sustained: argv[3] launches back to back (the way a prover issues them), ten at a time between two events -- does
			// the rate of the isolated launches above hold once the chip has been busy for a while?
			const int reps = atoi(argv[3]);
			for (int blk = 0; blk < reps / 10; blk++) {
				(void)hipEventRecord(ea);
				for (int r = 0; r < 10; r++) (void)launch_foldeval_mfma(0, 256, fa, n, z, d_out, nullptr);
				(void)hipEventRecord(eb);
				(void)hipEventSynchronize(eb);
				float ms;
				(void)hipEventElapsedTime(&ms, ea, eb);
				printf("fused N=2^%d sustained, launches %d..%d: %.3f ms each  %.2f TB/s\n", log_n, 10 * blk, 10 * blk + 9, ms / 10, n * 48.0 / (ms / 10) * 1e-9);
			}
		}
	}
	printf("%s\n", bad ? "FAILED" : "ALL OK");
	return bad != 0;
}
