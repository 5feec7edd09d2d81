// tools/mfma_round_phases.hip -- where the time of a MID-SIZE fused launch (k_foldeval_mfma, csrc/kernels_foldeval_mfma.hip:
// r = 19 ... 23, 512 ... 8192 tiles on 512 workgroups) goes: thread 0 of workgroup 0 stamps the 100 MHz wall clock at the phase
// boundaries (BN_TS), the host adds the launch -> mailbox round trip seen from its side.  Same method as
// tools/two_round_phases.hip.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude tools/mfma_round_phases.hip -o tools/mfma_round_phases
#define BN_PHASE_TS 1
#include "../binius_amd/csrc/kernels_foldeval_mfma.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bn;

__global__ void k_fill_rnd(f128 *p, uint64_t n, uint64_t seed)
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

int main()
{
	const int reps = 100;
	f128 *d_a, *d_b, *d_S, *d_rets, *mail;
	unsigned *d_counter;
	const uint64_t n_max = 1ull << 23;
	(void)hipMalloc(&d_a, n_max * 16);
	(void)hipMalloc(&d_b, n_max * 16);
	hipLaunchKernelGGL(k_fill_rnd, dim3(2048), dim3(256), 0, 0, d_a, n_max, 1);
	hipLaunchKernelGGL(k_fill_rnd, dim3(2048), dim3(256), 0, 0, d_b, n_max, 2);
	(void)hipMalloc(&d_S, 64 * 16);
	(void)hipMemset(d_S, 0, 64 * 16);
	(void)hipMalloc(&d_rets, 8 * 16);
	(void)hipMalloc(&d_counter, 4);
	(void)hipMemset(d_counter, 0, 4);
	(void)hipHostMalloc(&mail, 128 * 16, hipHostMallocCoherent | hipHostMallocMapped);
	for (int i = 0; i < 128; i++) mail[i] = f128{0, 0};
	hipStream_t s;
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	(void)hipDeviceSynchronize();
	// stamps: 0 entry, 1 loads issued, 2 table built + args committed, 3 first tile folded and staged, 4 tile loop done, 5 last Gram tile,
	// 6 parities -> three GF(2^64) sums per product, 7 recombined, 8 atomics + ticket, (9 .. 11 inside the finalize, last workgroup only), 12 exit
	static const char *names[13] = {"entry", "loads issued", "nibble table + args", "first tile (fold, stage)", "remaining tiles", "last Gram tile",
	                               "parity -> GF(2^64) sums", "recombine", "atomics + ticket", "", "", "", "to exit"};
	uint64_t seq = 0;
	for (uint64_t log_in : {19ull, 20ull, 21ull, 22ull, 23ull}) {
		const uint64_t n_in = 1ull << log_in;
		double acc[13] = {0};
		std::vector<double> host_all;
		for (int r = 0; r < reps + 3; r++) {
			foldeval_args fa{};
			fa.x0[0] = d_a;
			fa.x1[0] = d_a + n_in / 2;
			fa.out[0] = d_a;
			fa.x0[1] = d_b;
			fa.x1[1] = d_b + n_in / 2;
			fa.out[1] = d_b;
			fin_fuse fz{};
			fz.args.n_terms = fz.args.n_values = fz.args.n_ret = fz.args.n_slots = 2;
			fz.args.seq = ++seq;
			for (uint32_t t = 0; t < 2; t++) {
				fz.args.terms[t] = fin_term{t, t, f128{1, 0}};
				fz.args.ret_ids[t] = t;
			}
			fz.S = d_S;
			fz.rets = d_rets;
			fz.mail = mail;
			fz.counter = d_counter;
			const f128 z{0x1234567890abcdefull + r, 0xfedcba0987654321ull};
			const auto t0 = std::chrono::steady_clock::now();
			hipError_t e = launch_foldeval_mfma(s, 256, fa, n_in, z, d_S, &fz, nullptr);
			if (e != hipSuccess) {
				printf("launch failed: %s\n", hipGetErrorString(e));
				return 1;
			}
			while (__atomic_load_n(&mail[64].lo, __ATOMIC_ACQUIRE) != seq) {
			}
			const auto t1 = std::chrono::steady_clock::now();
			(void)hipStreamSynchronize(s);
			uint64_t ts[16];
			(void)hipMemcpyFromSymbol(ts, HIP_SYMBOL(bn_phase_ts), sizeof(ts));
			if (r >= 3) {
				host_all.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
				for (int i : {1, 2, 3, 4, 5, 6, 7, 8}) acc[i] += (double)(ts[i] - ts[i - 1]) * 0.01; // 100 MHz -> us
				acc[12] += (double)(ts[12] - ts[8]) * 0.01;
			}
		}
		std::sort(host_all.begin(), host_all.end());
		printf("k_foldeval_mfma<0>, n_in = 2^%llu (%llu tiles): host launch -> mailbox median %.2f us; in-kernel (workgroup 0):\n", (unsigned long long)log_in,
		       (unsigned long long)(n_in / 4 / 256), host_all[host_all.size() / 2]);
		double tot = 0;
		for (int i : {1, 2, 3, 4, 5, 6, 7, 8, 12}) {
			printf("  %-28s %6.2f us\n", names[i], acc[i] / reps);
			tot += acc[i] / reps;
		}
		printf("  %-28s %6.2f us\n", "sum", tot);
	}
	return 0;
}
