// tools/mfma_gram.hip -- prototype + rate measurement of the Gram-matrix formulation of the
// bivariate round evaluation:
//     S = sum_j u_j * v_j  (GF(2^128))  =  sum_{p,q} G[p][q] e_p e_q ,   G = U^T V  over GF(2)
// with G computed on the matrix cores (v_mfma_i32_32x32x32_i8): the operand bytes are the raw data
// bits left in place (one AND with a per-lane mask), the i32 accumulators count modulo 2^32 and only
// the parity bit is read back.  One Karatsuba level (three 64x64 Gram matrices per product).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibinius_amd/csrc -Iinclude tools/mfma_gram.hip -o /tmp/mfma_gram
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gf128.hpp"

using namespace bn;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {
constexpr int kTP = 512;                 // points per tile
constexpr int kBlkW = 40;                // words per (k-step, limb) block: 32 + 8 pad
constexpr int kSetW = 4 * 16 * kBlkW;    // words per operand set per tile: 4 k-slices x 16 blocks
constexpr int kBufW = 4 * kSetW;         // 4 operand sets
} // namespace

__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// 4 words (one per point) -> 4 words (one per byte index), byte k of output c = byte c of input k
__device__ __forceinline__ void btr4(const uint32_t (&r)[4], uint32_t (&o)[4])
{
	const uint32_t t01l = perm(r[1], r[0], 0x05010400u), t01h = perm(r[1], r[0], 0x07030602u);
	const uint32_t t23l = perm(r[3], r[2], 0x05010400u), t23h = perm(r[3], r[2], 0x07030602u);
	o[0] = perm(t23l, t01l, 0x05040100u);
	o[1] = perm(t23l, t01l, 0x07060302u);
	o[2] = perm(t23h, t01h, 0x05040100u);
	o[3] = perm(t23h, t01h, 0x07060302u);
}

__device__ __forceinline__ uint64_t mul_basis64(uint64_t z, unsigned i)
{
	if (i & 1) z = mulx64<0>(z);
	if (i & 2) z = mulx64<1>(z);
	if (i & 4) z = mulx64<2>(z);
	if (i & 8) z = mulx64<3>(z);
	if (i & 16) z = mulx64<4>(z);
	if (i & 32) z = mulx64<5>(z);
	return z;
}

__device__ __forceinline__ v4i and4(v4i x, uint32_t m)
{
	return v4i{(int)((uint32_t)x.x & m), (int)((uint32_t)x.y & m), (int)((uint32_t)x.z & m), (int)((uint32_t)x.w & m)};
}

__device__ __forceinline__ v4i bitop4(v4i x, v4i y, uint32_t m) // (x ^ y) & m
{
	return v4i{(int)__builtin_amdgcn_bitop3_b32((uint32_t)x.x, (uint32_t)y.x, m, 0x28), (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.y, (uint32_t)y.y, m, 0x28),
	           (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.z, (uint32_t)y.z, m, 0x28), (int)__builtin_amdgcn_bitop3_b32((uint32_t)x.w, (uint32_t)y.w, m, 0x28)};
}

// ORG 0 (X): no Karatsuba; wave (pr, rh, ksl): rows 64 rh .. 64 rh + 63 of the 128x128 Gram matrix, 8 tiles
// ORG 1 (Y): one Karatsuba level; wave (pr, h, ksl): column half h of the three 64x64 matrices, 6 tiles
// ORG 2 (Z): one Karatsuba level; wave (pr, q): all three 64x64 matrices over k-slice q, 12 tiles
template <int ORG, int MODE>
__global__ __launch_bounds__(512) void k_gram(const uint32_t *__restrict__ a_hi, const uint32_t *__restrict__ a_lo,
                                              const uint32_t *__restrict__ b_hi, const uint32_t *__restrict__ b_lo, uint64_t n,
                                              unsigned long long *out)
{
	extern __shared__ uint32_t lds[]; // [2][kBufW]
	const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	// ---- staging role: operand o (a / b), k-slice q, word column w, point group pg
	const unsigned o = wv >> 2, q = wv & 3, w = lane & 3, pg = lane >> 2;
	const uint32_t *p_hi = o ? b_hi : a_hi, *p_lo = o ? b_lo : a_lo;
	const unsigned st_off = (q * 16 + (pg >> 2) * 4 + w) * kBlkW + (pg & 3) * 2; // + c*8
	// ---- compute role
	const unsigned pr = wv >> 2, h = (wv >> 1) & 1, ksl = wv & 1;
	const unsigned m = lane & 31, kb = lane >> 5;
	const uint32_t msk = 0x01010101u << (m & 7);
	constexpr int NKS = ORG >= 2 ? 4 : 8;       // k-steps per wave and tile
	constexpr int NT = ORG == 0 ? 8 : (ORG == 1 ? 6 : (ORG == 2 ? 12 : 9));
	const unsigned rd_off = (ORG >= 2 ? q * 16 : ksl * 32) * kBlkW + (m >> 3) * 8 + kb * 4; // + (kstep*4 + w)*kBlkW

	v16i acc[NT];
#pragma unroll
	for (int s = 0; s < NT; s++)
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[s][r] = 0;

	const uint64_t n_tiles = n / kTP;
	uint32_t hi[8], lo[8];
	auto load = [&](uint64_t t) {
		const uint64_t base = (t * kTP + q * 128 + pg) * 4 + w;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			hi[i] = p_hi[base + 64 * i];
			lo[i] = p_lo[base + 64 * i];
		}
	};
	auto store = [&](unsigned buf) {
		uint32_t *dst_hi = lds + buf * kBufW + o * kSetW + st_off;
		uint32_t *dst_mx = dst_hi + 2 * kSetW;
		uint32_t th[2][4], tm[2][4];
#pragma unroll
		for (int g = 0; g < 2; g++) {
			const uint32_t rh[4] = {hi[4 * g], hi[4 * g + 1], hi[4 * g + 2], hi[4 * g + 3]};
			const uint32_t rm[4] = {hi[4 * g] ^ lo[4 * g], hi[4 * g + 1] ^ lo[4 * g + 1], hi[4 * g + 2] ^ lo[4 * g + 2],
			                        hi[4 * g + 3] ^ lo[4 * g + 3]};
			btr4(rh, th[g]);
			btr4(rm, tm[g]);
		}
#pragma unroll
		for (int c = 0; c < 4; c++) {
			*reinterpret_cast<uint2 *>(dst_hi + c * 8) = uint2{th[0][c], th[1][c]};
			*reinterpret_cast<uint2 *>(dst_mx + c * 8) = uint2{tm[0][c], tm[1][c]};
		}
	};
	auto compute = [&](unsigned buf) {
		const uint32_t *U = lds + buf * kBufW + (2 * pr) * kSetW + rd_off;
		const uint32_t *V = U + kSetW;
		v4i u[4], v[4];
		auto rd = [&](int ks) {
#pragma unroll
			for (int ww = 0; ww < 4; ww++) {
				if (ORG != 0 || (ww >> 1) == 0) // X: rows 64 rh..: limbs 2 rh, 2 rh + 1
					u[ww] = *reinterpret_cast<const v4i *>(U + (ks * 4 + (ORG == 0 ? 2 * h + ww : ww)) * kBlkW);
				if (ORG != 1 || (ww & 1) == 0) // Y: column half h: limbs h, 2 + h
					v[ww] = *reinterpret_cast<const v4i *>(V + (ks * 4 + (ORG == 1 ? ww + h : ww)) * kBlkW);
			}
		};
		if (MODE >= 3) rd(0);
#pragma unroll
		for (int ks = 0; ks < NKS; ks++) {
			if (MODE < 3) {
				rd(ks);
			} else {
				u[0].x += ks;
				v[0].y ^= ks;
			}
			if (ORG == 0) {
				v4i A[2], B;
				A[0] = and4(u[0], msk);
				A[1] = and4(u[1], msk);
#pragma unroll
				for (int j = 0; j < 4; j++) {
					B = and4(v[j], msk);
					acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0], B, acc[j], 0, 0, 0);
					acc[4 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1], B, acc[4 + j], 0, 0, 0);
				}
			} else if (ORG == 1) {
#pragma unroll
				for (int s = 0; s < 3; s++) {
					const v4i B = s == 0 ? and4(v[0], msk) : (s == 1 ? and4(v[2], msk) : bitop4(v[0], v[2], msk));
#pragma unroll
					for (int i = 0; i < 2; i++) {
						const v4i A = s == 0 ? and4(u[i], msk) : (s == 1 ? and4(u[2 + i], msk) : bitop4(u[i], u[2 + i], msk));
						acc[s * 2 + i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[s * 2 + i], 0, 0, 0);
					}
				}
			} else if (ORG == 3) {
				// two Karatsuba levels: 9 combinations of the four 32-bit limbs, one 32x32 tile each
#pragma unroll
				for (int c9 = 0; c9 < 9; c9++) {
					v4i A, B;
					switch (c9) {
					case 0: A = and4(u[0], msk); B = and4(v[0], msk); break;
					case 1: A = and4(u[1], msk); B = and4(v[1], msk); break;
					case 2: A = bitop4(u[0], u[1], msk); B = bitop4(v[0], v[1], msk); break;
					case 3: A = and4(u[2], msk); B = and4(v[2], msk); break;
					case 4: A = and4(u[3], msk); B = and4(v[3], msk); break;
					case 5: A = bitop4(u[2], u[3], msk); B = bitop4(v[2], v[3], msk); break;
					case 6: A = bitop4(u[0], u[2], msk); B = bitop4(v[0], v[2], msk); break;
					case 7: A = bitop4(u[1], u[3], msk); B = bitop4(v[1], v[3], msk); break;
					default: A = bitop4(u[0] ^ u[1] ^ u[2], u[3], msk); B = bitop4(v[0] ^ v[1] ^ v[2], v[3], msk); break;
					}
					acc[c9] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[c9], 0, 0, 0);
				}
			} else {
#pragma unroll
				for (int s = 0; s < 3; s++) {
					v4i A[2], B[2];
#pragma unroll
					for (int i = 0; i < 2; i++) {
						A[i] = s == 0 ? and4(u[i], msk) : (s == 1 ? and4(u[2 + i], msk) : bitop4(u[i], u[2 + i], msk));
						B[i] = s == 0 ? and4(v[i], msk) : (s == 1 ? and4(v[2 + i], msk) : bitop4(v[i], v[2 + i], msk));
					}
#pragma unroll
					for (int i = 0; i < 2; i++)
#pragma unroll
						for (int j = 0; j < 2; j++)
							acc[(s * 2 + i) * 2 + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i], B[j], acc[(s * 2 + i) * 2 + j], 0, 0, 0);
				}
			}
		}
	};

	uint64_t t = blockIdx.x;
	unsigned buf = 0;
	if (t < n_tiles) {
		load(t);
		store(0);
	}
	__syncthreads();
	for (; t < n_tiles; t += gridDim.x) {
		const uint64_t tn = t + gridDim.x;
		if (MODE < 1 && tn < n_tiles) load(tn);
		compute(buf);
		if (MODE < 2 && tn < n_tiles) store(buf ^ 1);
		__syncthreads();
		buf ^= 1;
	}

	// ---- parity bits of the accumulators -> Gram rows in LDS: Gw[product][sub][row][col word], XOR over k-slices
	// C[mm][nn]: nn = lane & 31, mm = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the parity sits at bit
	// (mm & 7) + (nn & 7) of the accumulator.
	uint32_t *Gw = lds; // [2][3][64][2] (ORG 0: [2][1][128][4])
	__syncthreads();
	for (unsigned i = tid; i < 2048; i += 512) Gw[i] = 0;
	__syncthreads();
#pragma unroll
	for (int tt = 0; tt < NT; tt++) {
		// tile tt of this wave: sub-matrix s, row tile i (32 rows), column tile j (32 columns)
		unsigned s, i, j;
		if (ORG == 3) { s = tt; i = 0; j = 0; }
		else if (ORG == 0) { s = 0; i = 2 * h + (tt >> 2); j = tt & 3; }
		else if (ORG == 1) { s = tt >> 1; i = tt & 1; j = h; }
		else { s = tt >> 2; i = (tt >> 1) & 1; j = tt & 1; }
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const unsigned sh = (r & 3) + 4 * kb + (lane & 7);
			const bool bit = ((uint32_t)acc[tt][r] >> sh) & 1u;
			const unsigned long long bal = __ballot(bit);
			const unsigned row0 = 32 * i + (r & 3) + 8 * (r >> 2);
			if (lane == 0) {
				if (ORG == 3) {
					atomicXor(&Gw[(pr * 9 + s) * 32 + row0], (uint32_t)bal);
					atomicXor(&Gw[(pr * 9 + s) * 32 + row0 + 4], (uint32_t)(bal >> 32));
				} else if (ORG == 0) {
					atomicXor(&Gw[(pr * 128 + row0) * 4 + j], (uint32_t)bal);
					atomicXor(&Gw[(pr * 128 + row0 + 4) * 4 + j], (uint32_t)(bal >> 32));
				} else {
					atomicXor(&Gw[((pr * 3 + s) * 64 + row0) * 2 + j], (uint32_t)bal);
					atomicXor(&Gw[((pr * 3 + s) * 64 + row0 + 4) * 2 + j], (uint32_t)(bal >> 32));
				}
			}
		}
	}
	__syncthreads();
	if (ORG == 0) {
		__shared__ f128 zz[2][2];
		if (wv < 4) {
			const unsigned p2 = wv >> 1, row = (wv & 1) * 64 + lane;
			const uint32_t *g = Gw + (p2 * 128 + row) * 4;
			f128 z = mul_basis(f128{(uint64_t)g[0] | ((uint64_t)g[1] << 32), (uint64_t)g[2] | ((uint64_t)g[3] << 32)}, row);
#pragma unroll
			for (int mm = 32; mm >= 1; mm >>= 1) {
				z.lo ^= __shfl_xor(z.lo, mm, 64);
				z.hi ^= __shfl_xor(z.hi, mm, 64);
			}
			if (lane == 0) zz[p2][wv & 1] = z;
		}
		__syncthreads();
		if (tid < 2) {
			const f128 z = zz[tid][0] ^ zz[tid][1];
			atomicXor(out + 2 * tid, (unsigned long long)z.lo);
			atomicXor(out + 2 * tid + 1, (unsigned long long)z.hi);
		}
		return;
	}
	if (ORG == 3) {
		__shared__ uint32_t z9[2][9];
		if (lane < 32) {
			for (unsigned k = wv; k < 18; k += 8) {
				uint32_t z = (uint32_t)mul_basis64((uint64_t)Gw[k * 32 + lane], lane);
#pragma unroll
				for (int mm = 16; mm >= 1; mm >>= 1)
					z ^= __shfl_xor(z, mm, 64);
				if (lane == 0) z9[k / 9][k % 9] = z;
			}
		}
		__syncthreads();
		if (tid < 2) {
			const uint32_t *pc = z9[tid];
			auto c32 = [](uint32_t z0, uint32_t z2, uint32_t z1p) {
				const uint32_t lo = z0 ^ z2;
				const uint32_t hi2 = z1p ^ lo ^ (uint32_t)mulx64<4>((uint64_t)z2);
				return (uint64_t)lo | ((uint64_t)hi2 << 32);
			};
			const uint64_t Z0 = c32(pc[0], pc[1], pc[2]), Z2 = c32(pc[3], pc[4], pc[5]), Z1 = c32(pc[6], pc[7], pc[8]);
			const uint64_t l = Z0 ^ Z2;
			const uint64_t hh = Z1 ^ l ^ mulx64<5>(Z2);
			atomicXor(out + 2 * tid, (unsigned long long)l);
			atomicXor(out + 2 * tid + 1, (unsigned long long)hh);
		}
		return;
	}
	__shared__ uint64_t zs[2][3];
	if (wv < 6) {
		const unsigned p2 = wv / 3, s = wv % 3; // product, sub-product; lane = row
		const uint32_t *g = Gw + ((p2 * 3 + s) * 64 + lane) * 2;
		uint64_t z = mul_basis64((uint64_t)g[0] | ((uint64_t)g[1] << 32), lane);
#pragma unroll
		for (int mm = 32; mm >= 1; mm >>= 1)
			z ^= __shfl_xor(z, mm, 64);
		if (lane == 0) zs[p2][s] = z;
	}
	__syncthreads();
	if (tid < 2) {
		const uint64_t Z0 = zs[tid][0], Z2 = zs[tid][1], Z1 = zs[tid][2];
		const uint64_t l = Z0 ^ Z2;
		const uint64_t hh = Z1 ^ l ^ mulx64<5>(Z2);
		atomicXor(out + 2 * tid, (unsigned long long)l);
		atomicXor(out + 2 * tid + 1, (unsigned long long)hh);
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
	const int log_n = argc > 1 ? atoi(argv[1]) : 20; // points (each array has n elements)
	const int log_check = argc > 2 ? atoi(argv[2]) : 14;
	const uint64_t n = 1ull << log_n;
	std::vector<f128> h[4];
	uint64_t seed = 1234;
	f128 *d[4];
	for (int k = 0; k < 4; k++) {
		h[k].resize(n);
		for (uint64_t i = 0; i < n; i++) h[k][i] = f128{sm64(seed), sm64(seed)};
		hipMalloc(&d[k], n * 16);
		hipMemcpy(d[k], h[k].data(), n * 16, hipMemcpyHostToDevice);
	}
	unsigned long long *d_out;
	hipMalloc(&d_out, 32);
	const size_t lds_bytes = 2 * kBufW * 4;
		auto check = [&](auto kern, const char *name) {
		hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		const uint64_t nc = 1ull << log_check;
		hipMemset(d_out, 0, 32);
		hipLaunchKernelGGL(kern, dim3(3), dim3(512), lds_bytes, 0, (const uint32_t *)d[0], (const uint32_t *)d[1],
		                   (const uint32_t *)d[2], (const uint32_t *)d[3], nc, d_out);
		f128 got[2];
		hipMemcpy(got, d_out, 32, hipMemcpyDeviceToHost);
		static f128 e1, ei;
		static bool have = false;
		if (!have) {
			e1 = f128_zero(); ei = f128_zero();
			for (uint64_t i = 0; i < nc; i++) {
				e1 ^= mul_slow(h[0][i], h[2][i]);
				ei ^= mul_slow(h[0][i] ^ h[1][i], h[2][i] ^ h[3][i]);
			}
			have = true;
		}
		printf("check %s n=2^%d: S1 %s  Sinf %s\n", name, log_check, (got[0] == e1) ? "OK" : "MISMATCH", (got[1] == ei) ? "OK" : "MISMATCH");
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) printf("hip error: %s\n", hipGetErrorString(e));
	};
	check(k_gram<0, 0>, "X");
	check(k_gram<1, 0>, "Y");
	check(k_gram<2, 0>, "Z");
	check(k_gram<3, 0>, "L2");
	hipEvent_t ea, eb;
	hipEventCreate(&ea);
	hipEventCreate(&eb);
	auto run = [&](auto kern, const char *name, int grid) {
		hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
		for (int rep = 0; rep < 2; rep++) {
			hipMemset(d_out, 0, 32);
			hipEventRecord(ea);
			hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, 0, (const uint32_t *)d[0], (const uint32_t *)d[1],
			                   (const uint32_t *)d[2], (const uint32_t *)d[3], n, d_out);
			hipEventRecord(eb);
			hipEventSynchronize(eb);
			float ms;
			hipEventElapsedTime(&ms, ea, eb);
			printf("%-22s n=2^%d grid %d: %.3f ms  %.2f G points/s  %.2f TB/s algorithmic (64 B/point)\n", name, log_n, grid, ms, n / ms * 1e-6,
			       n * 64.0 / ms * 1e-9);
		}
	};
	run(k_gram<0, 0>, "X full", 256);
	run(k_gram<0, 1>, "X no global loads", 256);
	run(k_gram<0, 2>, "X no loads/staging", 256);
	run(k_gram<0, 3>, "X mfma+valu only", 256);
	run(k_gram<1, 0>, "Y full", 256);
	run(k_gram<1, 1>, "Y no global loads", 256);
	run(k_gram<1, 2>, "Y no loads/staging", 256);
	run(k_gram<1, 3>, "Y mfma+valu only", 256);
	run(k_gram<3, 0>, "L2 full", 256);
	run(k_gram<3, 1>, "L2 no global loads", 256);
	run(k_gram<3, 2>, "L2 no loads/staging", 256);
	run(k_gram<3, 3>, "L2 mfma+valu only", 256);
	run(k_gram<2, 1>, "Z no global loads", 256);
	run(k_gram<2, 2>, "Z no loads/staging", 256);
	run(k_gram<2, 3>, "Z mfma+valu only", 256);
	return 0;
}
