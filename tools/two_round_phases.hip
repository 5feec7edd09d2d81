// tools/two_round_phases.hip -- where the time of a two-round launch (k_foldeval8, csrc/kernels_foldeval8.hip) goes: thread 0
// of workgroup 0 stamps the 100 MHz wall clock at the phase boundaries (BN_TS), the host adds the launch -> mailbox round
// trip seen from its side.  Same method as tools/small_round_phases.hip (the one-round kernel).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude tools/two_round_phases.hip -o tools/two_round_phases
#define BN_PHASE_TS 1
#include "../binius_amd/csrc/kernels_foldeval8.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bn;

int main()
{
	const int reps = 300;
	f128 *d_a, *d_b, *d_S, *d_rets, *mail;
	unsigned *d_counter;
	const uint64_t n_max = 1ull << 18;
	(void)hipMalloc(&d_a, n_max * 16);
	(void)hipMalloc(&d_b, n_max * 16);
	(void)hipMemset(d_a, 0x5a, n_max * 16);
	(void)hipMemset(d_b, 0x3c, n_max * 16);
	(void)hipMalloc(&d_S, 64 * 16);
	(void)hipMemset(d_S, 0, 64 * 16);
	(void)hipMalloc(&d_rets, 8 * 16);
	(void)hipMalloc(&d_counter, 4);
	(void)hipMemset(d_counter, 0, 4);
	(void)hipHostMalloc(&mail, 128 * 16, hipHostMallocCoherent | hipHostMallocMapped);
	for (int i = 0; i < 128; i++) mail[i] = f128{0, 0};
	hipStream_t s;
	(void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	static const char *names[9] = {"entry", "loads issued", "nibble tables built", "folds + stage", "rows read", "transposes + combos", "bs_mul", "collapse + recombine",
	                              "atomics + finalize"};
	uint64_t seq = 0;
	for (uint32_t nf : {2u, 1u}) {
		for (uint64_t n_in : {64ull, 1024ull, 16384ull, 65536ull, 131072ull}) {
			double acc[9] = {0};
			std::vector<double> host_all;
			for (int r = 0; r < reps + 3; r++) {
				foldeval8_args fa{};
				fa.x0[0] = d_a;
				fa.x1[0] = d_a + n_in / 2;
				fa.out[0] = d_a;
				fa.x0[1] = d_b;
				fa.x1[1] = d_b + n_in / 2;
				fa.out[1] = d_b;
				fa.n_in = n_in;
				fa.n_folds = nf;
				fin_fuse fz{};
				fz.args.n_terms = fz.args.n_values = fz.args.n_ret = fz.args.n_slots = 8;
				fz.args.seq = ++seq;
				for (uint32_t t = 0; t < 8; t++) {
					fz.args.terms[t] = fin_term{t, t, f128{1, 0}};
					fz.args.ret_ids[t] = t;
				}
				fz.S = d_S;
				fz.rets = d_rets;
				fz.mail = mail;
				fz.counter = d_counter;
				const f128 z1{0x1234567890abcdefull + r, 0xfedcba0987654321ull}, z2{0x0f1e2d3c4b5a6978ull ^ r, 0x1122334455667788ull};
				const auto t0 = std::chrono::steady_clock::now();
				hipError_t e = launch_foldeval8(s, fa, z1, z2, d_S, &fz, nullptr);
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
					for (int i = 1; i < 9; i++) acc[i] += (double)(ts[i] - ts[i - 1]) * 0.01; // 100 MHz -> us
				}
			}
			std::sort(host_all.begin(), host_all.end());
			printf("k_foldeval8<%u>, n_in = %llu (%llu workgroups): host launch -> mailbox median %.2f us; in-kernel (workgroup 0):\n", nf, (unsigned long long)n_in,
			       (unsigned long long)(((n_in >> nf) / 4 + 63) / 64), host_all[host_all.size() / 2]);
			double tot = 0;
			for (int i = 1; i < 9; i++) {
				printf("  %-24s %6.2f us\n", names[i], acc[i] / reps);
				tot += acc[i] / reps;
			}
			printf("  %-24s %6.2f us\n", "sum", tot);
		}
	}
	return 0;
}
