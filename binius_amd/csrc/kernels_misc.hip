// binius_amd/csrc/kernels_misc.hip -- the remaining ComputeLayerExecutor ops:
// inner_product / fold_left / fold_right (subfield x F, crates/compute/src/layer.rs:263,321,351),
// element-wise products (pairwise_product_reduce, layer.rs:505) and fri_fold (layer.rs:389).
//
// Subfield operands: a SubfieldSlice{slice, tower_level=iota} (crates/compute/src/memory.rs:257)
// is the same bytes read as 2^(7-iota) limbs of 2^iota bits per 16-byte element, least-significant
// limb first (iter_bases, crates/field/src/binary_field.rs:633-649).  limb * F is a bilinear walk
// over only 2^iota bits (gf128.hpp mul_walk<iota>), so small-field matrices cost almost nothing.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ctable.hpp"
#include "internal.hpp"

namespace bn {

__device__ __forceinline__ uint32_t wave_xor32(uint32_t v)
{
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1)
		v ^= __shfl_xor(v, m, 64);
	return v;
}

// XOR-reduce a per-thread f128 over a 256-thread block and atomically XOR it into out[0].
__device__ __forceinline__ void block_xor_to(f128 acc, f128 *out)
{
	__shared__ uint64_t red[4][2];
	uint32_t w[4] = {(uint32_t)acc.lo, (uint32_t)(acc.lo >> 32), (uint32_t)acc.hi, (uint32_t)(acc.hi >> 32)};
#pragma unroll
	for (int t = 0; t < 4; t++)
		w[t] = wave_xor32(w[t]);
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) {
		red[wave][0] = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
		red[wave][1] = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
	}
	__syncthreads();
	if (threadIdx.x < 2) {
		uint64_t v = red[0][threadIdx.x] ^ red[1][threadIdx.x] ^ red[2][threadIdx.x] ^ red[3][threadIdx.x];
		if (v)
			atomicXor(reinterpret_cast<unsigned long long *>(out) + threadIdx.x, (unsigned long long)v);
	}
}

// limb j (2^IOTA bits) of a packed subfield array
template <int IOTA>
__device__ __forceinline__ uint64_t subfield_limb(const uint64_t *words, uint64_t j)
{
	if constexpr (IOTA >= 6) {
		return words[j];
	} else {
		constexpr unsigned W = 1u << IOTA;
		constexpr unsigned PER = 64 / W;
		const uint64_t word = words[j / PER];
		return (word >> ((j % PER) * W)) & ((1ull << W) - 1);
	}
}

template <int IOTA>
__device__ __forceinline__ f128 mul_sub(f128 x, const uint64_t *words, uint64_t j)
{
	if constexpr (IOTA == 7) {
		return mul_slow(x, f128{words[2 * j], words[2 * j + 1]});
	} else {
		return mul_walk<IOTA>(x, subfield_limb<IOTA>(words, j));
	}
}

// inner_product: sum_i b[i] * a_sub[i]
template <int IOTA>
__global__ __launch_bounds__(256) void k_inner_product(const uint64_t *a, const uint4 *b, uint64_t n, f128 *out)
{
	f128 acc = f128_zero();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		acc ^= mul_sub<IOTA>(to_f128(b[i]), a, i);
	block_xor_to(acc, out);
}

// fold_right: out[i] = sum_j vec[j] * M[i*|vec| + j];  fold_left: out[i] = sum_j vec[j] * M[j*rows + i]
template <int IOTA, bool LEFT>
__global__ __launch_bounds__(256) void k_fold(const uint64_t *mat, const uint4 *vec, uint64_t vec_len, uint4 *out,
                                              uint64_t out_len)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < out_len; i += (uint64_t)gridDim.x * 256) {
		f128 acc = f128_zero();
		for (uint64_t j = 0; j < vec_len; j++) {
			const uint64_t idx = LEFT ? (j * out_len + i) : (i * vec_len + j);
			acc ^= mul_sub<IOTA>(to_f128(vec[j]), mat, idx);
		}
		out[i] = to_u4(acc);
	}
}

// fold_left / fold_right with nibble tables of the vector entries (IOTA in {3,4,5}).
// out[i] = sum_j vec[j] * m_ij with m_ij in T_IOTA: vec[j] is constant for the whole launch, so
// vec[j] * m = XOR_p T_j[p][nibble_p(m)], T_j[p][e] = vec[j] * (e << 4p), p < 2^IOTA/4.  A chunk of
// J = 256/P vectors' tables (64 KiB) lives in LDS; lookups are conflict-free for the same reason as
// in ctable.hpp (one 16-entry table = one bank row, all lanes use the same (j, p)).
// TH threads per workgroup: the 80 KiB of tables allow ONE workgroup per CU, so the workgroup itself has to
// bring the waves (256 threads = one wave per SIMD ran this kernel at a quarter of its rate)
template <int IOTA, bool LEFT, int R, int TH>
__global__ __launch_bounds__(TH) void k_fold_tab(const uint64_t *mat, const uint4 *vec, uint64_t vec_len, uint4 *out, uint64_t out_len)
{
	constexpr int P = (1 << IOTA) / 4;  // nibbles per subfield element
	constexpr int J = 256 / P;          // vectors per LDS chunk (J * P * 256 B = 64 KiB)
	constexpr int NB = 1 << IOTA;       // basis products per vector
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_fold[];
	uint4 *T = reinterpret_cast<uint4 *>(smem_fold);                 // [J][P][16]
	uint4 *basis = reinterpret_cast<uint4 *>(smem_fold + 65536);     // [J][NB]
	const unsigned tid = threadIdx.x;
	const uint64_t tile0 = (uint64_t)blockIdx.x * TH * R;
	uint4 acc[R];
#pragma unroll
	for (int r = 0; r < R; r++) acc[r] = uint4{0, 0, 0, 0};
	for (uint64_t j0 = 0; j0 < vec_len; j0 += J) {
		const uint64_t jn = (vec_len - j0) < (uint64_t)J ? (vec_len - j0) : (uint64_t)J;
		__syncthreads();
		for (unsigned q = tid; q < jn * NB; q += TH) {
			const unsigned jj = q / NB, b = q % NB;
			basis[q] = to_u4(mul_basis(to_f128(vec[j0 + jj]), b));
		}
		__syncthreads();
		for (unsigned q = tid; q < jn * P * 16; q += TH) {
			const unsigned e = q & 15, p = (q >> 4) % P, jj = q / (16 * P);
			const uint4 *bp = basis + jj * NB + 4 * p;
			uint4 v{0, 0, 0, 0};
			if (e & 1) v = xor4(v, bp[0]);
			if (e & 2) v = xor4(v, bp[1]);
			if (e & 4) v = xor4(v, bp[2]);
			if (e & 8) v = xor4(v, bp[3]);
			T[q] = v;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < R; r++) {
			const uint64_t i = tile0 + (uint64_t)r * TH + tid;
			if (i >= out_len) continue;
			// the matrix entries of a group are loaded before any of them is used: a load -> 8 dependent
			// lookups -> next load chain exposes the full memory latency once per entry at 2 waves per SIMD
			constexpr int G = J < 32 ? J : 32;
			for (uint64_t g0 = 0; g0 < jn; g0 += G) {
				uint32_t mm[G];
				constexpr int EB = (1 << IOTA) / 8;  // bytes per matrix entry
				constexpr int NU4 = G * EB / 16;      // 16-byte words per full group of a row
				if (!LEFT && NU4 >= 1 && vec_len % G == 0 && g0 + G <= jn) {
					// fold_right, full group: the G entries are contiguous in the thread's own row -- wide loads in
					// one burst (every 64-byte sector fetched once) instead of G strided narrow ones
					const uint4 *row = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(mat) + (i * vec_len + j0 + g0) * EB);
					uint32_t w[NU4 >= 1 ? 4 * NU4 : 4];
#pragma unroll
					for (int u = 0; u < NU4; u++) {
						const uint4 v = row[u];
						w[4 * u] = v.x;
						w[4 * u + 1] = v.y;
						w[4 * u + 2] = v.z;
						w[4 * u + 3] = v.w;
					}
#pragma unroll
					for (int k = 0; k < G; k++) {
						constexpr int PER = 4 / EB; // entries per 32-bit word
						mm[k] = IOTA == 5 ? w[k] : ((w[k / PER] >> ((k % PER) * 8 * EB)) & ((1u << (8 * EB)) - 1u));
					}
				} else {
					// entry-sized loads (fold_left: consecutive lanes read consecutive entries of row j)
					using entry_t = std::conditional_t<IOTA == 5, uint32_t, std::conditional_t<IOTA == 4, uint16_t, uint8_t>>;
					const entry_t *ent = reinterpret_cast<const entry_t *>(mat);
#pragma unroll
					for (int k = 0; k < G; k++) {
						const uint64_t jj = g0 + k;
						const uint64_t idx = LEFT ? ((j0 + jj) * out_len + i) : (i * vec_len + j0 + jj);
						mm[k] = jj < jn ? (uint32_t)ent[idx] : 0u;
					}
				}
				// lookups in groups of 16, all issued before any is consumed (otherwise every ds_read_b128 is
				// followed by its own s_waitcnt: one LDS round trip per lookup)
				constexpr int KG = 16 / P > 0 ? 16 / P : 1; // entries per lookup group
#pragma unroll
				for (int k = 0; k < G; k += KG) {
					if (g0 + k >= jn) break;
					uint4 t[KG * P];
#pragma unroll
					for (int kk = 0; kk < KG; kk++) {
						const bool on = g0 + k + kk < jn; // (beyond the chunk: mm = 0 -> entry 0 of table 0 = vec * 0 = 0)
						const char *tb = reinterpret_cast<const char *>(T + (on ? (g0 + k + kk) : 0) * P * 16);
						// nibble * 16 of every nibble in one instruction each (ctable.hpp: a rotate by four puts the even nibbles
						// where the odd ones are, an SDWA byte-select masks the high nibble of a byte in place)
						const uint32_t hi = mm[k + kk], lo = __builtin_amdgcn_alignbit(hi, hi, 28);
						uint32_t off[8];
						off[0] = byte_and<0>(lo, 0xF0u);
						off[1] = byte_and<0>(hi, 0xF0u);
						off[2] = byte_and<1>(lo, 0xF0u);
						off[3] = byte_and<1>(hi, 0xF0u);
						off[4] = byte_and<2>(lo, 0xF0u);
						off[5] = byte_and<2>(hi, 0xF0u);
						off[6] = byte_and<3>(lo, 0xF0u);
						off[7] = byte_and<3>(hi, 0xF0u);
#pragma unroll
						for (int p = 0; p < P; p++) t[kk * P + p] = *reinterpret_cast<const uint4 *>(tb + p * 256 + off[p]);
					}
					__builtin_amdgcn_sched_barrier(0);
					// (KG * P = 16 lookups, folded two at a time with the three-input XOR)
#pragma unroll
					for (int q = 0; q < KG * P; q += 2) {
						acc[r].x = ct_xor3(acc[r].x, t[q].x, t[q + 1].x);
						acc[r].y = ct_xor3(acc[r].y, t[q].y, t[q + 1].y);
						acc[r].z = ct_xor3(acc[r].z, t[q].z, t[q + 1].z);
						acc[r].w = ct_xor3(acc[r].w, t[q].w, t[q + 1].w);
					}
				}
			}
		}
	}
#pragma unroll
	for (int r = 0; r < R; r++) {
		const uint64_t i = tile0 + (uint64_t)r * TH + tid;
		if (i < out_len) out[i] = acc[r];
	}
}

template <int IOTA, bool LEFT>
static hipError_t run_fold_tab(hipStream_t s, const void *mat, const void *vec, uint64_t vec_len, void *out, uint64_t out_len)
{
	constexpr int R = 1, TH = 1024;
	constexpr int NB = 1 << IOTA, P = NB / 4, J = 256 / P;
	const size_t lds = 65536 + (size_t)J * NB * 16;
	const uint64_t blocks = (out_len + TH * R - 1) / (TH * R);
	const hipError_t attr = func_lds_limit(reinterpret_cast<const void *>(&k_fold_tab<IOTA, LEFT, R, TH>), (int)lds);
	if (attr != hipSuccess) return attr;
	hipLaunchKernelGGL((k_fold_tab<IOTA, LEFT, R, TH>), dim3((unsigned)blocks), dim3(TH), lds, s, (const uint64_t *)mat, (const uint4 *)vec, vec_len,
	                   (uint4 *)out, out_len);
	return hipGetLastError();
}

template <template <int> class Launcher, typename... Args>
static hipError_t dispatch_level(uint32_t level, Args... args)
{
	switch (level) {
	case 0: return Launcher<0>::run(args...);
	case 3: return Launcher<3>::run(args...);
	case 4: return Launcher<4>::run(args...);
	case 5: return Launcher<5>::run(args...);
	case 6: return Launcher<6>::run(args...);
	case 7: return Launcher<7>::run(args...);
	default: return hipErrorInvalidValue;
	}
}

template <int IOTA>
struct ip_launcher {
	static hipError_t run(hipStream_t s, unsigned g, const void *a, const void *b, uint64_t n, f128 *out)
	{
		hipLaunchKernelGGL(k_inner_product<IOTA>, dim3(g), dim3(256), 0, s, (const uint64_t *)a, (const uint4 *)b, n, out);
		return hipGetLastError();
	}
};

hipError_t launch_inner_product(hipStream_t s, int n_cu, const void *a, uint32_t tower_level, const void *b,
                                uint64_t b_len, f128 *d_out)
{
	if (b_len == 0) return hipSuccess;
	uint64_t want = (b_len + 255) / 256;
	unsigned g = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
	return dispatch_level<ip_launcher>(tower_level, s, g, a, b, b_len, d_out);
}

template <int IOTA>
struct foldl_launcher {
	static hipError_t run(hipStream_t s, unsigned g, const void *m, const void *v, uint64_t vl, void *o, uint64_t ol)
	{
		hipLaunchKernelGGL((k_fold<IOTA, true>), dim3(g), dim3(256), 0, s, (const uint64_t *)m, (const uint4 *)v, vl,
		                   (uint4 *)o, ol);
		return hipGetLastError();
	}
};
template <int IOTA>
struct foldr_launcher {
	static hipError_t run(hipStream_t s, unsigned g, const void *m, const void *v, uint64_t vl, void *o, uint64_t ol)
	{
		hipLaunchKernelGGL((k_fold<IOTA, false>), dim3(g), dim3(256), 0, s, (const uint64_t *)m, (const uint4 *)v, vl,
		                   (uint4 *)o, ol);
		return hipGetLastError();
	}
};

hipError_t launch_fold_left(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec,
                            uint64_t vec_len, void *out, uint64_t out_len)
{
	if (out_len == 0) return hipSuccess;
	if (out_len >= 1024) {
		if (tower_level == 3) return run_fold_tab<3, true>(s, mat, vec, vec_len, out, out_len);
		if (tower_level == 4) return run_fold_tab<4, true>(s, mat, vec, vec_len, out, out_len);
		if (tower_level == 5) return run_fold_tab<5, true>(s, mat, vec, vec_len, out, out_len);
	}
	uint64_t want = (out_len + 255) / 256;
	unsigned g = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
	return dispatch_level<foldl_launcher>(tower_level, s, g, mat, vec, vec_len, out, out_len);
}

hipError_t launch_fold_right(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec,
                             uint64_t vec_len, void *out, uint64_t out_len)
{
	if (out_len == 0) return hipSuccess;
	if (out_len >= 1024) {
		if (tower_level == 3) return run_fold_tab<3, false>(s, mat, vec, vec_len, out, out_len);
		if (tower_level == 4) return run_fold_tab<4, false>(s, mat, vec, vec_len, out, out_len);
		if (tower_level == 5) return run_fold_tab<5, false>(s, mat, vec, vec_len, out, out_len);
	}
	uint64_t want = (out_len + 255) / 256;
	unsigned g = (unsigned)(want < (uint64_t)n_cu * 8 ? want : (uint64_t)n_cu * 8);
	return dispatch_level<foldr_launcher>(tower_level, s, g, mat, vec, vec_len, out, out_len);
}

// ---- fri_fold as halving passes ---------------------------------------------------------------
// Reference semantics: crates/compute/src/cpu/layer.rs:345-388.  Per output chunk the reference
// folds 2^b interleaved symbols with the b interleave challenges, then for every fold challenge
// does an inverse-NTT butterfly + line extrapolation on adjacent pairs.  Pair k of the WHOLE array
// in fold round with current log_len L uses twiddle get_subspace_eval(L, k) (k = chunk<<(ls-1)|off
// is simply the global pair index), so every challenge is one streaming pass that halves the array:
//   interleave pass:  out[k] = x[2k] + (x[2k+1] - x[2k]) * r
//   fold pass:        u = x[2k], v = x[2k+1]; v += u; u += v*t_k; out[k] = u + (v - u) * r
// with r launch-constant (LDS nibble tables) and t_k in the small twiddle field (mul_walk).
template <int TW>
__global__ __launch_bounds__(256) void k_fri_pass(const uint4 *in, uint4 *out, uint64_t n_out, f128 r, int ntt_pass,
                                                  const uint64_t *s_row, int n_bits)
{
	__shared__ ctable_smem tab;
	__shared__ uint64_t s_basis[64];
	ctable_build(tab, r);
	if (threadIdx.x < 64)
		s_basis[threadIdx.x] = (ntt_pass && (int)threadIdx.x < n_bits) ? s_row[threadIdx.x] : 0;
	__syncthreads();
	for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n_out; k += (uint64_t)gridDim.x * 256) {
		uint4 u = in[2 * k], v = in[2 * k + 1];
		if (ntt_pass) {
			uint64_t t = 0; // OnTheFlyTwiddleAccess::get: subset sum over the bits of k (twiddle.rs:141-168)
			for (int b = 0; b < n_bits; b++)
				if ((k >> b) & 1)
					t ^= s_basis[b];
			v = xor4(v, u);
			u = xor4(u, to_u4(mul_walk<TW>(to_f128(v), t)));
		}
		out[k] = xor4(u, ctable_mul(tab, xor4(u, v)));
	}
}

// C (2 or 3) consecutive challenges in ONE pass: a thread folds 2^C adjacent elements down to one, all levels in registers.
// The separate passes move (1 + 1/2)(1 + 1/2 + 1/4 + ...) = 3 N elements through HBM for N inputs; here a pass reads N and writes
// N / 2^C.  What made the first attempt at this lose (four 16-byte loads per lane at a 64-byte lane stride
// quarter the efficiency of every request) is taken out by staging: the workgroup loads its 256 * 2^C contiguous elements
// with fully coalesced 16-byte loads into LDS (rows padded to 2^C + 1 elements: conflict-free 128-bit reads at the lane stride)
// and each thread picks its run from there.  One nibble table per challenge (10 KiB each) stays resident; no barrier between
// the levels.  Same arithmetic, same order of operations per pair as k_fri_pass.
struct fri_level {
	f128 r;
	const uint64_t *s_row; // twiddle basis of this level's fold round (nullptr: an interleave challenge, no butterfly)
	int n_bits;
};
template <int C>
struct fri_levels {
	fri_level l[C];
};
// NTT = false: every level is an interleave challenge (no butterfly code in the kernel at all).
// NTT = true: levels with a twiddle basis do the inverse butterfly first.  The product by the per-pair twiddle t -- an element
// of T_5 (B32) or below, acting limb-wise on the four 32-bit limbs of v (binary_field.rs:361-412) -- goes through the 64 KiB
// GF(2^8) product table of the context, copied into LDS: Karatsuba over the tower (pairwise_recursive_arithmetic.rs:18-28)
// from T_5 down to bytes, the four limbs side by side in the four bytes of a register so that every XOR and every
// multiplication by X_2 of the recombination is one SWAR operation for all of them: 36 byte look-ups + ~170 instructions
// against the 1450 of the bilinear walk (mul_walk<5>), which was 38 of the 78 us the two butterfly passes of the
// benchmarked fold took -- and which, with its chain of mulx multiples, is what spilled 200+ registers in a three-level pass.
template <int K>
__device__ __forceinline__ uint32_t mulx32(uint32_t a)
{
	constexpr uint32_t M = (uint32_t)lo_half_mask<K>();
	constexpr int H = 1 << K;
	const uint32_t l0 = a & M, l1 = (a >> H) & M;
	if constexpr (K == 0)
		return l1 | ((l0 ^ l1) << 1);
	else
		return l1 | ((l0 ^ mulx32<K - 1>(l1)) << H);
}
// four bytes (one per limb) times the byte whose table row is `row`
__device__ __forceinline__ uint32_t pmul8(const uint8_t *row, uint32_t r)
{
	const uint32_t b0 = row[r & 0xFF], b1 = row[(r >> 8) & 0xFF], b2 = row[(r >> 16) & 0xFF], b3 = row[r >> 24];
	return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}
struct p16x4 {
	uint32_t l, h; // low / high bytes of four T_4 elements
};
__device__ __forceinline__ p16x4 pmul16(const uint8_t *m8, p16x4 a, uint32_t s_lo, uint32_t s_hi)
{
	const uint32_t z0 = pmul8(m8 + (s_lo << 8), a.l), z2 = pmul8(m8 + (s_hi << 8), a.h), z1 = pmul8(m8 + ((s_lo ^ s_hi) << 8), a.l ^ a.h);
	const uint32_t lo = z0 ^ z2;
	return p16x4{lo, z1 ^ lo ^ mulx32<2>(z2)}; // X_3^2 = X_3 X_2 + 1
}
__device__ __forceinline__ uint4 mul_tw32_tab(const uint8_t *m8, uint4 v, uint32_t t)
{
	// 4 x 4 byte transpose: r_i = byte i of every limb
	const uint32_t t0 = __builtin_amdgcn_perm(v.y, v.x, 0x05010400), t1 = __builtin_amdgcn_perm(v.y, v.x, 0x07030602);
	const uint32_t t2 = __builtin_amdgcn_perm(v.w, v.z, 0x05010400), t3 = __builtin_amdgcn_perm(v.w, v.z, 0x07030602);
	const p16x4 a0{__builtin_amdgcn_perm(t2, t0, 0x05040100), __builtin_amdgcn_perm(t2, t0, 0x07060302)};
	const p16x4 a1{__builtin_amdgcn_perm(t3, t1, 0x05040100), __builtin_amdgcn_perm(t3, t1, 0x07060302)};
	const uint32_t s0 = t & 0xFF, s1 = (t >> 8) & 0xFF, s2 = (t >> 16) & 0xFF, s3 = t >> 24;
	const p16x4 z0 = pmul16(m8, a0, s0, s1), z2 = pmul16(m8, a1, s2, s3), z1 = pmul16(m8, p16x4{a0.l ^ a1.l, a0.h ^ a1.h}, s0 ^ s2, s1 ^ s3);
	const p16x4 lo{z0.l ^ z2.l, z0.h ^ z2.h};
	// X_4^2 = X_4 X_3 + 1; X_3 (l + h X_3) = h + (l + h X_2) X_3
	const p16x4 hi{z1.l ^ lo.l ^ z2.h, z1.h ^ lo.h ^ z2.l ^ mulx32<2>(z2.h)};
	const uint32_t u0 = __builtin_amdgcn_perm(lo.h, lo.l, 0x05010400), u1 = __builtin_amdgcn_perm(lo.h, lo.l, 0x07030602);
	const uint32_t u2 = __builtin_amdgcn_perm(hi.h, hi.l, 0x05010400), u3 = __builtin_amdgcn_perm(hi.h, hi.l, 0x07030602);
	return uint4{__builtin_amdgcn_perm(u2, u0, 0x05040100), __builtin_amdgcn_perm(u2, u0, 0x07060302), __builtin_amdgcn_perm(u3, u1, 0x05040100),
	             __builtin_amdgcn_perm(u3, u1, 0x07060302)};
}

// NW waves per workgroup; with NTT the workgroup also holds the 64 KiB byte table (dynamic LDS), so it is made of eight waves.
template <int C, bool NTT, int NW>
__global__ __launch_bounds__(NW * 64, NTT ? 1 : 2) void k_fri_pass_multi(const uint4 *in, uint4 *out, uint64_t n_out, fri_levels<C> lv, const uint4 *d_mul8)
{
	constexpr int E = 1 << C, ROW = E + 1;
	__shared__ ctable_smem tab[C];
	__shared__ uint64_t s_basis[C][64];
	__shared__ uint4 stage[NW * 64 * ROW];
	extern __shared__ __attribute__((aligned(16))) uint8_t m8[]; // NTT: 65536 bytes
	{
		// the challenges' nibble tables side by side: groups of 128 threads build one table each (a build is a dependent chain
		// of ~400 instructions whatever the number of threads: one after the other they cost 1 us each at the head of a pass)
		constexpr int NGRP = NW * 64 / 128;
		const unsigned grp = threadIdx.x >> 7, ltid = threadIdx.x & 127;
#pragma unroll
		for (int c0 = 0; c0 < C; c0 += NGRP) {
			f128 r = lv.l[c0].r;
			unsigned mine = 0; // (compile-time indices only: a run-time index into the by-value argument makes every thread copy it to scratch)
#pragma unroll
			for (int g = 1; g < NGRP; g++)
				if (c0 + g < C && grp == (unsigned)g) {
					r = lv.l[c0 + g < C ? c0 + g : 0].r;
					mine = g;
				}
			const bool has = c0 + (int)grp < C;
			ctable_build_group(tab[has ? c0 + mine : c0], r, has ? ltid : 128u, 128u);
		}
	}
	if (threadIdx.x < 64) {
#pragma unroll
		for (int c = 0; c < C; c++)
			s_basis[c][threadIdx.x] = (lv.l[c].s_row && (int)threadIdx.x < lv.l[c].n_bits) ? lv.l[c].s_row[threadIdx.x] : 0;
	}
	if constexpr (NTT) {
		for (unsigned i = threadIdx.x; i < 4096; i += NW * 64) reinterpret_cast<uint4 *>(m8)[i] = d_mul8[i];
	}
	__syncthreads();
	// Every WAVE works on its own blocks of 64 outputs (64 * E contiguous inputs) with its own slice of the stage and no
	// workgroup barrier in the loop: the next block's inputs are requested (coalesced, into registers) before the current
	// block is multiplied, and the waves of a workgroup drift apart so that one's loads hide behind another's lookups.
	// (With a workgroup-wide stage and two barriers per block the counters showed the waves waiting 61 % of the time with the
	// VALU 40 % and the LDS 50 % busy: 134 us for the pass that the three passes it replaces also took.)
	const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint4 *wstage = stage + wave * 64 * ROW;
	const uint64_t n_wblocks = (n_out + 63) / 64, n_in = n_out << C;
	const uint64_t wstride = (uint64_t)gridDim.x * NW;
	uint64_t b = (uint64_t)blockIdx.x * NW + wave;
	uint4 pre[E];
	auto fetch = [&](uint64_t blk) {
		const uint64_t in0 = (blk * 64) << C;
#pragma unroll
		for (int j = 0; j < E; j++) {
			const uint64_t e = in0 + (uint64_t)j * 64 + lane; // coalesced
			pre[j] = e < n_in ? in[e] : uint4{0, 0, 0, 0};
		}
	};
	if (b < n_wblocks) fetch(b);
	for (; b < n_wblocks; b += wstride) {
		// registers -> this wave's stage (element e of the block at row e >> C, column e & (E - 1))
#pragma unroll
		for (int j = 0; j < E; j++) {
			const unsigned e = j * 64 + lane;
			wstage[(e >> C) * ROW + (e & (E - 1))] = pre[j];
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const uint64_t k = b * 64 + lane;
		{
			uint4 x[E];
			const uint4 *row = wstage + lane * ROW;
#pragma unroll
			for (int j = 0; j < E; j++) x[j] = row[j];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier(); // (every lane has its row in registers: the stage may be refilled)
			if (b + wstride < n_wblocks) fetch(b + wstride); // flies while this block is multiplied
#pragma unroll
			for (int c = 0; c < C; c++) {
				const int cnt = E >> (c + 1); // outputs of this level per thread
#pragma unroll
				for (int j = 0; j < E / 2; j++) {
					if (j < cnt) {
						uint4 u = x[2 * j], v = x[2 * j + 1];
						if (NTT && lv.l[c].s_row) { // (uniform)
							const uint64_t kk = (k << (C - 1 - c)) + j; // pair index in the whole array at this level
							uint64_t t = 0;
							for (int bb = 0; bb < lv.l[c].n_bits; bb++)
								if ((kk >> bb) & 1) t ^= s_basis[c][bb];
							v = xor4(v, u);
							u = xor4(u, mul_tw32_tab(m8, v, (uint32_t)t));
						}
						x[j] = xor4(u, ctable_mul<NTT ? 8 : 16>(tab[c], xor4(u, v)));
						// two pairs at a time in the interleave-only passes, one at a time next to the butterflies: left alone
						// the scheduler hoists the lookups of ALL pairs of a level to the front and spills hundreds of registers
						// 
						if (NTT || (j & 1) || j + 1 == cnt) __builtin_amdgcn_sched_barrier(0);
					}
				}
			}
			if (k < n_out) out[k] = x[0];
		}
	}
}

// (twiddles of at most 32 bits: B8 / B16 / B32 NTT fields)
hipError_t run_fri_multi(hipStream_t s, int n_cu, int C, const uint4 *src, uint4 *dst, uint64_t n_out, const fri_level *lv, const uint8_t *d_mul8)
{
	bool ntt = false;
	for (int c = 0; c < C; c++) ntt = ntt || lv[c].s_row != nullptr;
	const uint64_t n_wblocks = (n_out + 63) / 64;
	if (ntt) {
		{
			hipError_t e = func_lds_limit(reinterpret_cast<const void *>(&k_fri_pass_multi<2, true, 8>), 65536);
			if (e != hipSuccess) return e;
			e = func_lds_limit(reinterpret_cast<const void *>(&k_fri_pass_multi<3, true, 4>), 65536);
			if (e != hipSuccess) return e;
		}
		// one workgroup per CU (the byte table)
		if (C == 3) {
			const uint64_t want = (n_wblocks + 3) / 4;
			const unsigned g = (unsigned)(want < (uint64_t)n_cu ? want : (uint64_t)n_cu);
			fri_levels<3> a{{lv[0], lv[1], lv[2]}};
			hipLaunchKernelGGL((k_fri_pass_multi<3, true, 4>), dim3(g), dim3(256), 65536, s, src, dst, n_out, a, (const uint4 *)d_mul8);
		} else {
			const uint64_t want = (n_wblocks + 7) / 8;
			const unsigned g = (unsigned)(want < (uint64_t)n_cu ? want : (uint64_t)n_cu);
			fri_levels<2> a{{lv[0], lv[1]}};
			hipLaunchKernelGGL((k_fri_pass_multi<2, true, 8>), dim3(g), dim3(512), 65536, s, src, dst, n_out, a, (const uint4 *)d_mul8);
		}
	} else {
		const uint64_t want = (n_wblocks + 3) / 4;
		const unsigned g = (unsigned)(want < (uint64_t)n_cu * 2 ? want : (uint64_t)n_cu * 2); // two workgroups per CU (62 KiB of LDS each)
		if (C == 3) {
			fri_levels<3> a{{lv[0], lv[1], lv[2]}};
			hipLaunchKernelGGL((k_fri_pass_multi<3, false, 4>), dim3(g), dim3(256), 0, s, src, dst, n_out, a, nullptr);
		} else {
			fri_levels<2> a{{lv[0], lv[1]}};
			hipLaunchKernelGGL((k_fri_pass_multi<2, false, 4>), dim3(g), dim3(256), 0, s, src, dst, n_out, a, nullptr);
		}
	}
	return hipGetLastError();
}

hipError_t launch_fri_fold(hipStream_t s, const uint64_t *d_s_evals, uint32_t tw_level, uint32_t log_domain,
                           uint32_t log_len, uint32_t log_batch, const f128 *h_challenges, uint32_t n_challenges,
                           const void *in, void *out, uint64_t out_len, void *scratch, int n_cu, const uint8_t *d_mul8)
{
	// scratch holds two ping-pong buffers of in_len/2 elements each
	const uint64_t in_len = out_len << n_challenges;
	uint4 *buf0 = (uint4 *)scratch;
	uint4 *buf1 = buf0 + in_len / 2;
	const uint4 *src = (const uint4 *)in;
	uint64_t cur = in_len;
	uint32_t ll = log_len;
	static const bool multi = [] {
		const char *e = bn::settled_knob("BN_FRI_MULTI");
		return !(e && e[0] == '0');
	}();
	static const bool ntt_c3 = [] {
		const char *e = bn::settled_knob("BN_FRI_NTT_C3"); // measurement knob: three levels per pass next to butterflies too
		return e && e[0] == '1';
	}();
	uint32_t c = 0, pass = 0;
	while (c < n_challenges) {
		// challenges per pass: three at a time (then two, then one) while the pass is large enough to be a streaming pass
		// (three interleave challenges per pass; two as soon as a butterfly level is among them -- the general kernel with
		// three levels spills -- and one at a time for B64 twiddles)
		uint32_t C = 1;
		if (multi && cur >= (1u << 14)) {
			const uint32_t left = n_challenges - c, inter = c < log_batch ? log_batch - c : 0;
			if (inter >= 2)
				C = inter >= 3 ? 3 : 2;
			else if (tw_level <= 5 && left >= 2)
				C = (ntt_c3 && left >= 3) ? 3 : 2;
		}
		const uint64_t n_out = cur >> C;
		uint4 *dst = (c + C == n_challenges) ? (uint4 *)out : ((pass & 1) ? buf1 : buf0);
		fri_level lv[3];
		for (uint32_t q = 0; q < C; q++) {
			const bool ntt_pass = c + q >= log_batch;
			lv[q].r = h_challenges[c + q];
			// twiddles of get_subspace_eval(ll, .) = s_evals[log_domain - ll], which has log_domain-1-(log_domain-ll)=ll-1 bits
			lv[q].s_row = ntt_pass ? d_s_evals + (uint64_t)(log_domain - ll) * BN_NTT_MAX_DIM : nullptr;
			lv[q].n_bits = ntt_pass ? (int)ll - 1 : 0;
			if (ntt_pass) ll -= 1;
		}
		hipError_t e;
		if (C == 1) {
			uint64_t want = (n_out + 255) / 256;
			unsigned g = (unsigned)(want < 2048 ? want : 2048);
			if (g < 1) g = 1;
			const int ntt_pass = lv[0].s_row != nullptr;
			const uint64_t *row = ntt_pass ? lv[0].s_row : d_s_evals;
			switch (tw_level) {
			case 3: hipLaunchKernelGGL(k_fri_pass<3>, dim3(g), dim3(256), 0, s, src, dst, n_out, lv[0].r, ntt_pass, row, lv[0].n_bits); break;
			case 4: hipLaunchKernelGGL(k_fri_pass<4>, dim3(g), dim3(256), 0, s, src, dst, n_out, lv[0].r, ntt_pass, row, lv[0].n_bits); break;
			case 5: hipLaunchKernelGGL(k_fri_pass<5>, dim3(g), dim3(256), 0, s, src, dst, n_out, lv[0].r, ntt_pass, row, lv[0].n_bits); break;
			case 6: hipLaunchKernelGGL(k_fri_pass<6>, dim3(g), dim3(256), 0, s, src, dst, n_out, lv[0].r, ntt_pass, row, lv[0].n_bits); break;
			default: return hipErrorInvalidValue;
			}
			e = hipGetLastError();
		} else {
			if (tw_level < 3 || tw_level > 6) return hipErrorInvalidValue;
			e = run_fri_multi(s, n_cu, (int)C, src, dst, n_out, lv, d_mul8);
		}
		if (e != hipSuccess) return e;
		src = dst;
		cur = n_out;
		c += C;
		pass++;
	}
	return hipSuccess;
}

// ---- is a table the tensor expansion of some coordinates (up to a constant)? -----------------------------------------
// eq[i] == eq[i - 2^k] * rho_k for every i in [1, n), k = the top bit of i, rho_k = eq[2^k] / eq[0] (host).  One pass over
// the table: workgroups are dealt to the ranges [2^k, 2^(k+1)) (a.first_wg), each builds the nibble table of its rho_k;
// indices below 256 are checked by workgroup 0 with the generic product.  Any mismatch sets *flag.  Used once per MLE-check
// by the weighted shadow of abi_kernels.cpp (the literal BivariateMLEcheckProver call sequence, v3/bivariate_mlecheck.rs).
// rho[k] = eq[2^k] / eq[0] (host) and the workgroup layout are read from memory (indexing a by-value argument array with a
// run-time index makes every thread copy the array to scratch).  Four entries in
// flight per lane: one at a time the loop is a chain of HBM
// round trips.  (169 registers: three workgroups per CU.)
__global__ __launch_bounds__(256, 2) void k_check_tensor(const uint4 *eq, uint64_t n, const f128 *rho, const uint32_t *first_wg, uint32_t n_log, unsigned *flag)
{
	__shared__ ctable_smem tab;
	__shared__ uint32_t s_k, s_first, s_next;
	if (threadIdx.x == 0) {
		unsigned k = 8;
		while (k + 1 < n_log && blockIdx.x >= first_wg[k + 1]) k++;
		s_k = k;
		s_first = first_wg[k];
		s_next = first_wg[k + 1];
	}
	__syncthreads();
	const unsigned k = s_k;
	const bool ranged = n_log > 8;
	ctable_build(tab, rho[k < n_log ? k : 0]);
	bool bad = false;
	if (ranged) {
		const uint64_t lo = (uint64_t)1 << k, hi = lo << 1;
		const uint64_t part = blockIdx.x - s_first, n_wg = s_next - s_first;
		for (uint64_t i0 = lo + part * 1024 + threadIdx.x; i0 < hi && i0 < n; i0 += n_wg * 1024) {
			uint4 p[4], g[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint64_t i = i0 + 256 * j;
				const bool in = i < hi && i < n;
				p[j] = in ? eq[i - lo] : uint4{0, 0, 0, 0};
				g[j] = in ? eq[i] : uint4{0, 0, 0, 0};
			}
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint4 want = ctable_mul<8>(tab, p[j]);
				bad = bad || want.x != g[j].x || want.y != g[j].y || want.z != g[j].z || want.w != g[j].w;
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
	// (the first 256 entries, k = 0 .. 7: k_check_tensor_head -- the generic product they need would cost this kernel its
	// occupancy: 288 registers, one wave per SIMD, 130 - 180 us for 2^23 entries)
	if (bad) atomicOr(flag, 1u);
}

// d_rho[n_log], d_first_wg[42]: device-visible copies of the ratios and of the workgroup prefix built by check_tensor_layout
uint32_t check_tensor_layout(uint32_t n_log, uint32_t *first_wg /*[42]*/)
{
	uint32_t wg = 0;
	for (uint32_t k = 0; k <= 41; k++) {
		first_wg[k] = wg;
		if (k >= 8 && k < n_log) {
			// a workgroup per 4096 elements, at most 512 per range (every workgroup builds a nibble table first: with one per
			// 1024 elements a fifth of the kernel was table building)
			uint64_t w = ((uint64_t)1 << k) / 4096;
			wg += (uint32_t)(w < 1 ? 1 : (w > 512 ? 512 : w));
		}
	}
	return wg ? wg : 1; // tables of at most 256 entries: workgroup 0 alone
}

// the first 256 entries (k = 0 .. 7) with the generic product, in a kernel of their own (inside k_check_tensor the product's
// registers cost the whole kernel its occupancy)
__global__ __launch_bounds__(256) void k_check_tensor_head(const uint4 *eq, uint64_t n, const f128 *rho, unsigned *flag)
{
	const uint64_t i = threadIdx.x;
	if (i >= 1 && i < n) {
		const unsigned kt = 31 - __clz((unsigned)i);
		const f128 want = mul_slow(to_f128(eq[i - ((uint64_t)1 << kt)]), rho[kt]);
		if (!(want == to_f128(eq[i]))) atomicOr(flag, 1u);
	}
}

hipError_t launch_check_tensor(hipStream_t s, const void *eq, uint64_t n, const f128 *d_rho, const uint32_t *d_first_wg, uint32_t n_wg, uint32_t n_log,
                               unsigned *d_flag)
{
	if (n_log > 40) return hipErrorNotSupported;
	hipLaunchKernelGGL(k_check_tensor_head, dim3(1), dim3(256), 0, s, (const uint4 *)eq, n, d_rho, d_flag);
	if (n_log <= 8) return hipGetLastError();
	hipLaunchKernelGGL(k_check_tensor, dim3(n_wg), dim3(256), 0, s, (const uint4 *)eq, n, d_rho, d_first_wg, n_log, d_flag);
	return hipGetLastError();
}

} // namespace bn
