// tools/small_round_phases.hip -- where the ~13 us of a small fused round (k_foldeval9_small) go: thread 0 of
// workgroup 0 stamps the 100 MHz wall clock at the phase boundaries (BN_TS in kernels_foldeval9.hip / re9.hpp);
// the host adds the launch -> mailbox round trip seen from its side.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude tools/small_round_phases.hip -o tools/small_round_phases
#define BN_PHASE_TS 1
#include "../binius_amd/csrc/kernels_foldeval9.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bn;

int main(int argc, char **argv)
{
	const int reps = 300;
	f128 *d_a, *d_b, *d_S, *d_rets, *mail;
	unsigned *d_counter;
	const uint64_t n_max = 1ull << 16;
	(void)hipMalloc(&d_a, n_max * 16);
	(void)hipMalloc(&d_b, n_max * 16);
	(void)hipMemset(d_a, 0x5a, n_max * 16);
	(void)hipMemset(d_b, 0x3c, n_max * 16);
	(void)hipMalloc(&d_S, 64 * 16);
	(void)hipMemset(d_S, 0, 64 * 16);
	(void)hipMalloc(&d_rets, 8 * 16);
	(void)hipMalloc(&d_counter, 4);
	(void)hipMemset(d_counter, 0, 4);
	(void)hipHostMalloc(&mail, 65 * 16, hipHostMallocCoherent | hipHostMallocMapped);
	for (int i = 0; i < 65; i++) mail[i] = f128{0, 0};
	hipStream_t s;
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	static const char *names[12] = {"entry", "loads issued", "ctable built", "fold + stage", "transposes + combos", "bs_mul", "collapse", "recombine",
	                               "atomics + finalize", "fin: args + sums in LDS", "fin: terms folded", "fin: stores issued"};
	uint64_t seq = 0;
	for (uint64_t n_in : {256ull, 2048ull, 16384ull, 65536ull}) {
		double acc[12] = {0};
		double host_us = 0;
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
			fz.args.n_terms = 2;
			fz.args.n_values = 2;
			fz.args.n_ret = 2;
			fz.args.n_slots = 2;
			fz.args.seq = ++seq;
			for (int t = 0; t < 2; t++) {
				fz.args.terms[t].slot = t;
				fz.args.terms[t].value = t;
				fz.args.terms[t].coeff = f128{1, 0};
				fz.args.ret_ids[t] = t;
			}
			fz.S = d_S;
			fz.rets = d_rets;
			fz.mail = mail;
			fz.counter = d_counter;
			const f128 z{0x1234567890abcdefull + r, 0xfedcba0987654321ull};
			const auto t0 = std::chrono::steady_clock::now();
			hipError_t e = launch_foldeval9(s, 256, fa, n_in, z, d_S, &fz);
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
				host_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
				host_all.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
				for (int i = 1; i < 9; i++) acc[i] += (double)(ts[i] - ts[i - 1]) * 0.01; // 100 MHz -> us
				// the finalize stamps lie inside phase 8 (single-workgroup rounds only)
				acc[9] += (double)(ts[9] - ts[7]) * 0.01;
				acc[10] += (double)(ts[10] - ts[9]) * 0.01;
				acc[11] += (double)(ts[11] - ts[10]) * 0.01;
			}
		}
		std::sort(host_all.begin(), host_all.end());
		printf("n_in = %llu (%s): host launch -> mailbox mean %.2f us, median %.2f us; in-kernel (workgroup 0):\n", (unsigned long long)n_in,
		       foldeval9_is_small(256, n_in) ? "k_foldeval9_small" : "k_foldeval9", host_us / reps, host_all[host_all.size() / 2]);
		double tot = 0;
		for (int i = 1; i < 9; i++) {
			printf("  %-22s %6.2f us\n", names[i], acc[i] / reps);
			tot += acc[i] / reps;
		}
		printf("  %-22s %6.2f us\n", "sum", tot);
		if (n_in <= 448)
			for (int i = 9; i < 12; i++) printf("    %-26s %6.2f us\n", names[i], acc[i] / reps);
	}
	return 0;
}
