// binius_amd/csrc/kernels_linmap.hip -- fold_right as a GF(2)-linear map on the matrix cores.
//
// fold_right (crates/compute/src/layer.rs:351, cpu/layer.rs `fold_right`): out[i] = sum_j mat[i][j] * vec[j], mat entries in a
// subfield of 2^IOTA bits, vec in GF(2^128).  A subfield scalar acts bit-wise -- m * v = sum_b m_b (2^b * v) -- so row i of the
// matrix, read as ROW_BITS = vec_len * 2^IOTA bits, selects which of ROW_BITS launch constants c_k = 2^(k mod 2^IOTA) * vec[k >> IOTA]
// are XORed into out[i]: a 128 x ROW_BITS bit matrix C applied to every row, i.e. an integer matrix product of which only the
// parity of every entry is wanted -- the same observation that put the round evaluations on the matrix cores (gram.hpp), with
// the roles changed: here the CONSTANT is one operand and the data the other, and the data needs no staging at all.
//
//  * v_mfma_scale_f32_32x32x64_f8f6f4, both operands FP4 (E2M1): K = 64 nibbles per instruction.  The B operand is the row
//    itself: lane (n, kh) -- n = row of a 32-row block, kh = K half -- loads the 16 bytes at offset 32 t + 16 kh of ITS row (32
//    nibbles, in place), and keeps one bit of every nibble with one AND: bit s = 0, 1, 2 decodes to 0.5, 1, 2; bit 3 is the sign
//    bit of the format, so it is moved to bit 2 first ((x >> 1) & 0x44444444: two instructions).
//  * The A operand holds C: entry (output bit r, nibble c, bit s) is the reciprocal code of what B's bit decodes to when bit r of
//    c_{4c+s} is set, zero otherwise -- every product is 1 and the f32 accumulator of (r, row) counts the selected constants
//    with bit r set (exact: at most 2048 terms).  A wave owns ONE 32-bit limb of the output (M tile w = output bits 32 w ..
//    32 w + 31) and keeps its slice of C in REGISTERS for the whole launch: ROW_BITS / 16 of them (128 at 2048 bits), loaded
//    once from a table a small kernel builds from vec.  (Round 3 measured this map with C in LDS -- 128 KiB, every MFMA paying
//    a 1 KiB LDS read: no faster than the nibble tables, DESIGN.md 4.13.  In registers the inner loop is one 16-byte global
//    load, 4 - 5 ANDs and 4 MFMAs per 32 rows and K-chunk.)
//  * Parity of the counts = the output bits; the two lane halves of a row are merged with one DPP-free shuffle and every wave
//    stores its limb of the 32 rows.
//
// 2^20 rows x 2048 bits: 4.2 M MFMAs = 4096 per SIMD.  tools/mfma_issue_fp4.hip puts this instruction at 18 - 20 ns back to back on
// random operands (two waves per SIMD) and at 21.7 ns with eight bitwise VALU instructions per MFMA, this kernel's mix: 74 and 89 us.
// Measured: 104 us + 5 us for the table (k_linmap_ring) against 134 us for the nibble-table kernel, which is bound by the LDS
// bandwidth of its 8 lookups per 32-bit entry.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ctable.hpp"
#include "internal.hpp"

namespace bn {

namespace {

typedef int lm_v8i __attribute__((ext_vector_type(8)));
typedef float lm_v16f __attribute__((ext_vector_type(16)));

// A table: [M tile w 0..3][step = 4 t + s][lane] uint4; lane (m, kh) holds the 32 K-entries (kh, reg, nib) of output bit 32 w + m.
// A workgroup builds the four steps of one (w, t): they read the 256 constants c_k, k = 256 t .. 256 t + 255, one per thread
// through LDS (a first version let every thread recompute its 32 constants: 11 us for a 128 KiB table).
template <int IOTA>
__global__ __launch_bounds__(256) void k_linmap_prep(const uint4 *__restrict__ vec, uint32_t steps, uint4 *__restrict__ A)
{
	__shared__ uint64_t c_lo[256], c_hi[256];
	const uint32_t T = steps >> 2;
	const uint32_t w = blockIdx.x / T, t = blockIdx.x % T;
	{
		const uint32_t bit = 256 * t + threadIdx.x; // bit index inside the row
		const f128 c = mul_basis(to_f128(vec[bit >> IOTA]), bit & ((1u << IOTA) - 1u));
		c_lo[threadIdx.x] = c.lo;
		c_hi[threadIdx.x] = c.hi;
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, s = threadIdx.x >> 6, m = lane & 31, kh = lane >> 5;
	const uint32_t r = 32 * w + m;
	const uint64_t *cw = r < 64 ? c_lo : c_hi;
	const uint32_t code = s == 0 ? 4u : (s == 1 ? 2u : 1u); // 2, 1, 0.5, 0.5: the reciprocal of what the data bit decodes to
	uint32_t regs[4] = {0, 0, 0, 0};
#pragma unroll
	for (uint32_t reg = 0; reg < 4; reg++)
#pragma unroll
		for (uint32_t nib = 0; nib < 8; nib++) {
			const uint32_t knib = 32 * kh + 8 * reg + nib; // nibble inside the chunk
			if ((cw[4 * knib + s] >> (r & 63)) & 1) regs[reg] |= code << (4 * nib);
		}
	A[((size_t)w * steps + 4 * t + s) * 64 + lane] = uint4{regs[0], regs[1], regs[2], regs[3]};
}

#define BN_LM_MFMA(ACC, A, B) BN_LM_MFMA_(ACC, A, B)
#define BN_LM_MFMA_(ACC, A, B)                                                                                                              \
	ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(lm_v8i{(int)(A).x, (int)(A).y, (int)(A).z, (int)(A).w, 0, 0, 0, 0},               \
	                                                      lm_v8i{(int)(B).x, (int)(B).y, (int)(B).z, (int)(B).w, 0, 0, 0, 0}, ACC, 4, 4, 0, \
	                                                      0x7F7F7F7F, 0, 0x7F7F7F7F)

__device__ __forceinline__ uint4 lm_bit(uint4 x, int s)
{
	if (s < 3) {
		const uint32_t m = 0x11111111u << s;
		return uint4{x.x & m, x.y & m, x.z & m, x.w & m};
	}
	return uint4{(x.x >> 1) & 0x44444444u, (x.y >> 1) & 0x44444444u, (x.z >> 1) & 0x44444444u, (x.w >> 1) & 0x44444444u};
}

// T = ROW_BITS / 256 chunks of 32 bytes per row, STEPS = 4 T MFMAs per 32 rows and wave
template <int T>
__global__ __launch_bounds__(256, 2) void k_linmap(const char *__restrict__ mat, const uint4 *__restrict__ A, uint32_t *__restrict__ out,
                                                   uint64_t n_rows)
{
	constexpr int STEPS = 4 * T;
	constexpr uint64_t ROW_BYTES = 32 * T;
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned n = lane & 31, kh = lane >> 5;
	uint4 a[STEPS];
#pragma unroll
	for (int st = 0; st < STEPS; st++)
		a[st] = A[((size_t)wave * STEPS + st) * 64 + lane];
	const uint64_t n_blocks = (n_rows + 31) / 32;
	auto row_ptr = [&](uint64_t blk) -> const uint4 * {
		uint64_t row = blk * 32 + n;
		if (row >= n_rows) row = n_rows - 1; // (its result is not stored)
		return reinterpret_cast<const uint4 *>(mat + row * ROW_BYTES + 16 * kh);
	};
	uint4 x[T], xn[T];
	uint64_t blk = blockIdx.x;
	if (blk >= n_blocks) return;
	{
		const uint4 *p = row_ptr(blk);
#pragma unroll
		for (int t = 0; t < T; t++)
			x[t] = p[2 * t];
	}
	for (; blk < n_blocks; blk += gridDim.x) {
		const uint64_t nb = blk + gridDim.x < n_blocks ? blk + gridDim.x : blk; // (the last block re-requests itself)
		{
			const uint4 *p = row_ptr(nb);
#pragma unroll
			for (int t = 0; t < T; t++)
				xn[t] = p[2 * t];
		}
		lm_v16f acc;
#pragma unroll
		for (int r = 0; r < 16; r++)
			acc[r] = 0.0f;
#pragma unroll
		for (int t = 0; t < T; t++) {
#pragma unroll
			for (int s = 0; s < 4; s++) {
				const uint4 b = lm_bit(x[t], s);
				BN_LM_MFMA(acc, a[4 * t + s], b);
			}
		}
		// register r of the tile: output bit (r & 3) + 8 (r >> 2) + 4 kh of this wave's limb, column = row n
		uint32_t part = 0;
#pragma unroll
		for (int r = 0; r < 16; r++)
			part |= ((uint32_t)(int)acc[r] & 1u) << ((r & 3) + 8 * (r >> 2));
		part <<= 4 * kh;
		const uint32_t word = part | (uint32_t)__shfl_xor((int)part, 32, 64);
		const uint64_t row = blk * 32 + n;
		if (kh == 0 && row < n_rows) out[row * 4 + wave] = word;
#pragma unroll
		for (int t = 0; t < T; t++)
			x[t] = xn[t];
	}
}

// The same map with the rows staged through an LDS ring by LDS-DMA loads (global_load_lds_dwordx4: no registers held while
// the data flies).  k_linmap above requests a block of rows one block ahead -- half a microsecond of MFMAs, a fraction of the
// memory latency -- and the waves spend most of their time waiting (0.186 ms at 2^20 x 2048 bits against 0.134 for the
// nibble tables); more register buffers do not fit beside 128 registers of C.  Here the four waves of a workgroup share ONE
// copy of every block (each fetches a quarter of it), D = 8 blocks deep.  A block of 32 rows is one contiguous stretch of
// memory, and every load instruction takes 1 KiB of it in order (lane i the i-th 16 bytes: fully coalesced -- a first version
// let every lane fetch from its own row, 32 cache lines per instruction, and ran at the texture path's pace: 0.129 ms); the
// instruction's kilobyte lands at its own LDS base, 1040 bytes after the previous one, which spreads the rows that a K half
// reads together over the banks (what is left is a four-way conflict between the rows of one instruction: a few cycles per
// block).  As in kernels_roundeval_fp4.hip the loads, their waits and the barrier are inline assembly: the compiler would
// wait for every outstanding LDS-DMA load before any LDS read it can see.  The wait is conservative -- at most (D - 2) x (loads
// per block and wave) operations outstanding, output stores included -- which leaves about four blocks in flight.
// ABL (measurement builds only): bit 0 = no MFMAs, bit 1 = no loads (the ring keeps whatever it holds), bit 2 = no barrier.
// At 2^20 rows x 2048 bits (profiles/r04/experiments/linmap_ablation.txt): 104 us as shipped; 49 without the MFMAs; 101 without the
// loads; 39 without both; 97 without loads and barrier: neither the memory system nor the barrier holds the kernel, the matrix
// pipe with its ~8 VALU instructions per MFMA does (89 us by tools/mfma_issue_fp4.hip, profiles/r04/experiments/mfma_issue_fp4.txt).
template <int T, int ABL = 0>
__global__ __launch_bounds__(256, 2) void k_linmap_ring(const char *__restrict__ mat, const uint4 *__restrict__ A, uint32_t *__restrict__ out,
                                                        uint64_t n_rows)
{
	// A step = a PAIR of 32-row blocks (64 consecutive rows): two independent accumulator chains per wave -- one chain alone
	// issues a dependent MFMA every ~30 ns, half the pipe's rate (measured: 0.118 ms with one chain) -- and one barrier per pair.
	constexpr int STEPS = 4 * T;
	constexpr int ROW_BYTES = 32 * T;
	constexpr int D = 4;                      // ring slots (pairs)
	constexpr int RPI = 1024 / ROW_BYTES;     // rows per load instruction
	constexpr int NI = 64 / RPI;              // load instructions per pair (= 2 T)
	constexpr int IPW = NI / 4;               // ... per wave (T = 2: one each)
	constexpr int ISTRIDE = 1040;             // LDS bytes between the kilobytes of two instructions
	constexpr int SLOT_BYTES = NI * ISTRIDE;
	extern __shared__ __attribute__((aligned(16))) unsigned char lm_ring[];
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned n = lane & 31, kh = lane >> 5;
	uint4 a[STEPS];
#pragma unroll
	for (int st = 0; st < STEPS; st++)
		a[st] = A[((size_t)wave * STEPS + st) * 64 + lane];
	const uint64_t n_pairs = n_rows / 64; // (the launcher sends whole pairs only)
	const uint64_t my_pairs = blockIdx.x < n_pairs ? (n_pairs - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
	if (my_pairs == 0) return;
	const uint32_t ring_base = (uint32_t)(uintptr_t)lm_ring;
	// pair number j of this workgroup into its slot; pairs past the end re-fetch the last one (the operation counts stay uniform)
	auto fetch = [&](uint64_t j) {
		const uint64_t jj = j < my_pairs ? j : my_pairs - 1;
		const uint64_t pr = blockIdx.x + jj * gridDim.x;
		const char *g0 = mat + pr * (64 * (uint64_t)ROW_BYTES) + lane * 16;
		const uint32_t slot = ring_base + (uint32_t)(j % D) * SLOT_BYTES;
#pragma unroll
		for (int u = 0; u < IPW; u++) {
			const unsigned k = wave + 4 * u; // instruction k of the pair: its k-th kilobyte
			const char *g = g0 + k * 1024;
			const uint32_t l0 = __builtin_amdgcn_readfirstlane(slot + k * ISTRIDE);
			if constexpr (!(ABL & 2)) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l0) : "memory", "m0");
		}
	};
	for (uint64_t j = 0; j + 1 < (uint64_t)D; j++) fetch(j);
	// row rr of the pair, chunk q = 2 t + kh: instruction rr / RPI, row rr % RPI inside its kilobyte
	const unsigned rd0 = (n / RPI) * ISTRIDE + (n % RPI) * ROW_BYTES + kh * 16;
	const unsigned rd1 = ((n + 32) / RPI) * ISTRIDE + ((n + 32) % RPI) * ROW_BYTES + kh * 16;
	for (uint64_t j = 0; j < my_pairs; j++) {
		// pair j has landed for this wave ... and, behind the barrier, for all four
		if constexpr ((ABL & 4) != 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
		else if constexpr ((D - 2) * IPW == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
		else if constexpr ((D - 2) * IPW == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
		else if constexpr ((D - 2) * IPW == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
		fetch(j + D - 1); // into the slot of pair j - 1: every wave is past it
		const unsigned char *slot = lm_ring + (j % D) * SLOT_BYTES;
		lm_v16f acc0, acc1;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			acc0[r] = 0.0f;
			acc1[r] = 0.0f;
		}
		// Half steps (t, h): the operands of bits 2 h, 2 h + 1 of both blocks (four uint4) and their four MFMAs.  The operands of
		// half step k + 1 are formed -- into the OTHER register set -- between the MFMAs of half step k: an operand register that
		// is rewritten right behind the MFMA that reads it makes the VALU wait for the matrix pipe, and the two ran one after the
		// other (0.110 ms).  The interleaving is pinned with scheduling groups (one MFMA, then five VALU).
		uint4 x0 = *reinterpret_cast<const uint4 *>(slot + rd0), x1 = *reinterpret_cast<const uint4 *>(slot + rd1);
		uint4 m[2][4]; // [register set][2 * (bit & 1) + block]
		m[0][0] = lm_bit(x0, 0);
		m[0][1] = lm_bit(x1, 0);
		m[0][2] = lm_bit(x0, 1);
		m[0][3] = lm_bit(x1, 1);
#pragma unroll
		for (int k = 0; k < 2 * T; k++) {
			const int t = k >> 1, h = k & 1, cur = k & 1, nxt = cur ^ 1;
			uint4 nx0 = x0, nx1 = x1;
			if (h == 1 && t + 1 < T) { // the next chunk of the two rows
				nx0 = *reinterpret_cast<const uint4 *>(slot + rd0 + (t + 1) * 32);
				nx1 = *reinterpret_cast<const uint4 *>(slot + rd1 + (t + 1) * 32);
			}
			if constexpr (!(ABL & 1)) {
				BN_LM_MFMA(acc0, a[4 * t + 2 * h], m[cur][0]);
				BN_LM_MFMA(acc1, a[4 * t + 2 * h], m[cur][1]);
				BN_LM_MFMA(acc0, a[4 * t + 2 * h + 1], m[cur][2]);
				BN_LM_MFMA(acc1, a[4 * t + 2 * h + 1], m[cur][3]);
			} else { // (keep the operands alive)
				acc0[k & 15] += __builtin_bit_cast(float, m[cur][0].x ^ m[cur][2].y ^ a[4 * t + 2 * h].x);
				acc1[k & 15] += __builtin_bit_cast(float, m[cur][1].z ^ m[cur][3].w ^ a[4 * t + 2 * h + 1].y);
			}
			if (k + 1 < 2 * T) {
				const int hn = h ^ 1;
				const uint4 &s0 = h == 1 ? nx0 : x0, &s1 = h == 1 ? nx1 : x1;
				m[nxt][0] = lm_bit(s0, 2 * hn);
				m[nxt][1] = lm_bit(s1, 2 * hn);
				m[nxt][2] = lm_bit(s0, 2 * hn + 1);
				m[nxt][3] = lm_bit(s1, 2 * hn + 1);
			}
#pragma unroll
			for (int g = 0; g < 4; g++) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
				__builtin_amdgcn_sched_group_barrier(0x002, 5, 0); // five VALU
			}
			x0 = nx0;
			x1 = nx1;
		}
		uint32_t p0 = 0, p1 = 0;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			p0 |= ((uint32_t)(int)acc0[r] & 1u) << ((r & 3) + 8 * (r >> 2));
			p1 |= ((uint32_t)(int)acc1[r] & 1u) << ((r & 3) + 8 * (r >> 2));
		}
		p0 <<= 4 * kh;
		p1 <<= 4 * kh;
		// lanes 0..31 take the other half of block 0's word, lanes 32..63 of block 1's: every lane then stores one limb
		const uint32_t give = kh ? p0 : p1, keep = kh ? p1 : p0;
		const uint32_t word = keep | (uint32_t)__shfl_xor((int)give, 32, 64);
		const uint64_t row = (blockIdx.x + j * gridDim.x) * 64 + lane;
		out[row * 4 + wave] = word;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (this wave's LDS reads of the slot are done before it reaches the next barrier)
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// fold_left (layer.rs:321): out[j] = sum_i mat[i * n_out + j] * vec[i] -- the same map with the "row" of output j gathered from
// the vec_len matrix rows.  B32 entries only (one dword per (i, j)): a load instruction brings the 64 entries (i, 64 p .. 64 p + 63)
// of a step -- 256 contiguous bytes, one dword per lane -- into LDS row i of the slot, and the B operand of (t, kh) is the four
// dwords of rows 8 t + 4 kh .. + 3 at the lane's column (four ds_read_b32: their bank pattern is the lane index, conflict-free).
// Bit k of the gathered row = bit k % 32 of entry i = k / 32: the K order of fold_right's rows, so the table is the same.
template <int T>
__global__ __launch_bounds__(256, 2) void k_linmap_ring_left(const uint32_t *__restrict__ mat, const uint4 *__restrict__ A, uint32_t *__restrict__ out,
                                                             uint64_t n_out)
{
	constexpr int STEPS = 4 * T;
	constexpr int L = 8 * T;       // matrix rows (= vec_len)
	constexpr int D = 4;           // ring slots (steps of 64 outputs)
	constexpr int IPW = L / 4;     // load instructions per step and wave
	constexpr int SLOT_BYTES = L * 256;
	extern __shared__ __attribute__((aligned(16))) unsigned char lm_ring[];
	const unsigned lane = threadIdx.x & 63;
	const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const unsigned n = lane & 31, kh = lane >> 5;
	uint4 a[STEPS];
#pragma unroll
	for (int st = 0; st < STEPS; st++)
		a[st] = A[((size_t)wave * STEPS + st) * 64 + lane];
	const uint64_t n_pairs = n_out / 64;
	const uint64_t my_pairs = blockIdx.x < n_pairs ? (n_pairs - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
	if (my_pairs == 0) return;
	const uint32_t ring_base = (uint32_t)(uintptr_t)lm_ring;
	auto fetch = [&](uint64_t j) {
		const uint64_t jj = j < my_pairs ? j : my_pairs - 1;
		const uint64_t pr = blockIdx.x + jj * gridDim.x;
		const uint32_t *g0 = mat + pr * 64 + lane;
		const uint32_t slot = ring_base + (uint32_t)(j % D) * SLOT_BYTES;
#pragma unroll
		for (int u = 0; u < IPW; u++) {
			const unsigned i = wave + 4 * u; // matrix row
			const uint32_t *g = g0 + (uint64_t)i * n_out;
			const uint32_t l0 = __builtin_amdgcn_readfirstlane(slot + i * 256);
			asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l0) : "memory", "m0");
		}
	};
	for (uint64_t j = 0; j + 1 < (uint64_t)D; j++) fetch(j);
	const unsigned rd0 = (4 * kh) * 256 + n * 4, rd1 = rd0 + 128;
	auto chunk = [&](const unsigned char *base, int t) -> uint4 {
		const unsigned char *q = base + t * 8 * 256;
		return uint4{*reinterpret_cast<const uint32_t *>(q), *reinterpret_cast<const uint32_t *>(q + 256), *reinterpret_cast<const uint32_t *>(q + 512),
		             *reinterpret_cast<const uint32_t *>(q + 768)};
	};
	for (uint64_t j = 0; j < my_pairs; j++) {
		static_assert((D - 2) * IPW <= 63, "vmcnt range");
		if constexpr ((D - 2) * IPW == 32) asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
		else if constexpr ((D - 2) * IPW == 16) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
		else if constexpr ((D - 2) * IPW == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
		fetch(j + D - 1);
		const unsigned char *slot = lm_ring + (j % D) * SLOT_BYTES;
		lm_v16f acc0, acc1;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			acc0[r] = 0.0f;
			acc1[r] = 0.0f;
		}
		uint4 x0 = chunk(slot + rd0, 0), x1 = chunk(slot + rd1, 0);
		uint4 m[2][4];
		m[0][0] = lm_bit(x0, 0);
		m[0][1] = lm_bit(x1, 0);
		m[0][2] = lm_bit(x0, 1);
		m[0][3] = lm_bit(x1, 1);
#pragma unroll
		for (int k = 0; k < 2 * T; k++) {
			const int t = k >> 1, h = k & 1, cur = k & 1, nxt = cur ^ 1;
			uint4 nx0 = x0, nx1 = x1;
			if (h == 1 && t + 1 < T) {
				nx0 = chunk(slot + rd0, t + 1);
				nx1 = chunk(slot + rd1, t + 1);
			}
			BN_LM_MFMA(acc0, a[4 * t + 2 * h], m[cur][0]);
			BN_LM_MFMA(acc1, a[4 * t + 2 * h], m[cur][1]);
			BN_LM_MFMA(acc0, a[4 * t + 2 * h + 1], m[cur][2]);
			BN_LM_MFMA(acc1, a[4 * t + 2 * h + 1], m[cur][3]);
			if (k + 1 < 2 * T) {
				const int hn = h ^ 1;
				const uint4 &s0 = h == 1 ? nx0 : x0, &s1 = h == 1 ? nx1 : x1;
				m[nxt][0] = lm_bit(s0, 2 * hn);
				m[nxt][1] = lm_bit(s1, 2 * hn);
				m[nxt][2] = lm_bit(s0, 2 * hn + 1);
				m[nxt][3] = lm_bit(s1, 2 * hn + 1);
			}
#pragma unroll
			for (int g = 0; g < 4; g++) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
				__builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
			}
			x0 = nx0;
			x1 = nx1;
		}
		uint32_t p0 = 0, p1 = 0;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			p0 |= ((uint32_t)(int)acc0[r] & 1u) << ((r & 3) + 8 * (r >> 2));
			p1 |= ((uint32_t)(int)acc1[r] & 1u) << ((r & 3) + 8 * (r >> 2));
		}
		p0 <<= 4 * kh;
		p1 <<= 4 * kh;
		const uint32_t give = kh ? p0 : p1, keep = kh ? p1 : p0;
		const uint32_t word = keep | (uint32_t)__shfl_xor((int)give, 32, 64);
		const uint64_t col = (blockIdx.x + j * gridDim.x) * 64 + lane;
		out[col * 4 + wave] = word;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int IOTA>
hipError_t linmap_prep(hipStream_t s, const void *vec, uint32_t steps, void *A)
{
	hipLaunchKernelGGL(k_linmap_prep<IOTA>, dim3(steps), dim3(256), 0, s, (const uint4 *)vec, steps, (uint4 *)A); // 4 M tiles x (steps / 4) chunks
	return hipGetLastError();
}

} // namespace

size_t linmap_table_bytes(uint64_t row_bits) { return (size_t)(row_bits / 64) * 4 * 64 * sizeof(uint4); }

// fold_right on the matrix cores for rows of 512, 1024 or 2048 bits; hipErrorNotSupported otherwise (the caller runs the
// nibble-table kernel).  d_table: linmap_table_bytes(row_bits) bytes of scratch.
hipError_t launch_fold_right_mfma(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec, uint64_t vec_len, void *out,
                                  uint64_t out_len, void *d_table)
{
	static const bool on = [] {
		const char *e = bn::settled_knob("BN_FOLD_MFMA"); // 0: the nibble-table kernel everywhere
		return !(e && e[0] == '0');
	}();
	const uint64_t row_bits = vec_len << tower_level;
	if (!on || !d_table || out_len < 4096 || (out_len & 63) || (row_bits != 512 && row_bits != 1024 && row_bits != 2048)) return hipErrorNotSupported;
	const uint32_t steps = (uint32_t)(row_bits / 64);
	hipError_t e;
	switch (tower_level) {
	case 0: e = linmap_prep<0>(s, vec, steps, d_table); break;
	case 3: e = linmap_prep<3>(s, vec, steps, d_table); break;
	case 4: e = linmap_prep<4>(s, vec, steps, d_table); break;
	case 5: e = linmap_prep<5>(s, vec, steps, d_table); break;
	case 6: e = linmap_prep<6>(s, vec, steps, d_table); break;
	case 7: e = linmap_prep<7>(s, vec, steps, d_table); break;
	default: return hipErrorNotSupported;
	}
	if (e != hipSuccess) return e;
	const uint64_t n_blocks = (out_len + 31) / 32;
	const uint64_t cap = (uint64_t)n_cu * 2;
	const dim3 grid((unsigned)(n_blocks < cap ? n_blocks : cap));
	static const bool ring = [] {
		const char *e = bn::settled_knob("BN_FOLD_MFMA_RING"); // 0: rows prefetched in registers, one block ahead (measurement knob)
		return !(e && e[0] == '0');
	}();
	if (ring) {
		const uint64_t n_pairs = out_len / 64;
		const dim3 grid((unsigned)(n_pairs < cap ? n_pairs : cap));
		const size_t lds = (size_t)4 * 2 * (row_bits / 256) * 1040; // D = 4 slots of 2 T instruction strides
		const hipError_t a2 = func_lds_limit(reinterpret_cast<const void *>(&k_linmap_ring<2>), 4 * 4 * 1040);
		const hipError_t a4 = func_lds_limit(reinterpret_cast<const void *>(&k_linmap_ring<4>), 4 * 8 * 1040);
		const hipError_t a8 = func_lds_limit(reinterpret_cast<const void *>(&k_linmap_ring<8>), 4 * 16 * 1040);
		if (a2 != hipSuccess) return a2;
		if (a4 != hipSuccess) return a4;
		if (a8 != hipSuccess) return a8;
		switch (row_bits) {
		case 512: hipLaunchKernelGGL(k_linmap_ring<2>, grid, dim3(256), lds, s, (const char *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
		case 1024: hipLaunchKernelGGL(k_linmap_ring<4>, grid, dim3(256), lds, s, (const char *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
		default: hipLaunchKernelGGL(k_linmap_ring<8>, grid, dim3(256), lds, s, (const char *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
		}
		return hipGetLastError();
	}
	switch (row_bits) {
	case 512: hipLaunchKernelGGL(k_linmap<2>, grid, dim3(256), 0, s, (const char *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
	case 1024: hipLaunchKernelGGL(k_linmap<4>, grid, dim3(256), 0, s, (const char *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
	default: hipLaunchKernelGGL(k_linmap<8>, grid, dim3(256), 0, s, (const char *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
	}
	return hipGetLastError();
}

// fold_left on the matrix cores: B32 entries, vec_len = 16, 32 or 64 (gathered rows of 512 / 1024 / 2048 bits)
hipError_t launch_fold_left_mfma(hipStream_t s, int n_cu, const void *mat, uint32_t tower_level, const void *vec, uint64_t vec_len, void *out,
                                 uint64_t out_len, void *d_table)
{
	static const bool on = [] {
		const char *e = bn::settled_knob("BN_FOLD_MFMA");
		return !(e && e[0] == '0');
	}();
	if (!on || !d_table || tower_level != 5 || out_len < 4096 || (out_len & 63) || (vec_len != 16 && vec_len != 32 && vec_len != 64)) return hipErrorNotSupported;
	const uint32_t steps = (uint32_t)(vec_len * 32 / 64);
	hipError_t e = linmap_prep<5>(s, vec, steps, d_table);
	if (e != hipSuccess) return e;
	const uint64_t n_pairs = out_len / 64, cap = (uint64_t)n_cu * 2;
	const dim3 grid((unsigned)(n_pairs < cap ? n_pairs : cap));
	const size_t lds = (size_t)4 * vec_len * 256;
	const hipError_t a2 = func_lds_limit(reinterpret_cast<const void *>(&k_linmap_ring_left<2>), 4 * 16 * 256);
	const hipError_t a4 = func_lds_limit(reinterpret_cast<const void *>(&k_linmap_ring_left<4>), 4 * 32 * 256);
	const hipError_t a8 = func_lds_limit(reinterpret_cast<const void *>(&k_linmap_ring_left<8>), 4 * 64 * 256);
	if (a2 != hipSuccess) return a2;
	if (a4 != hipSuccess) return a4;
	if (a8 != hipSuccess) return a8;
	switch (vec_len) {
	case 16: hipLaunchKernelGGL(k_linmap_ring_left<2>, grid, dim3(256), lds, s, (const uint32_t *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
	case 32: hipLaunchKernelGGL(k_linmap_ring_left<4>, grid, dim3(256), lds, s, (const uint32_t *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
	default: hipLaunchKernelGGL(k_linmap_ring_left<8>, grid, dim3(256), lds, s, (const uint32_t *)mat, (const uint4 *)d_table, (uint32_t *)out, out_len); break;
	}
	return hipGetLastError();
}

} // namespace bn
