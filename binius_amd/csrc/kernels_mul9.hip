// binius_amd/csrc/kernels_mul9.hip -- element-wise GF(2^128) products, bit-sliced:
//   out[i] = a[i * a_stride] * b[b_off + i * b_stride]
// used by compute_composite for product compositions (crates/compute/src/layer.rs:459),
// pairwise_product_reduce (layer.rs:505: a = in, strides 2, b_off 1) and as the first step of the
// MLE-check round evaluation (b * eq, then the bivariate kernel).
//
// Same wave mapping as kernels_roundeval9.hip -- 7 groups of 9 lanes, each lane owning one of the 9
// GF(2^32) limb-combination products of two Karatsuba levels -- but here the products are needed per
// element, so after the multiplication the 9 partial products are exchanged through LDS, four lanes
// per group rebuild the four 32-bit limbs of the result in the bit-sliced domain
//     R0 = p0+p1+p3+p4                         R1 = p0+..+p5 + α(p1+p4)
//     R2 = p0+p1+p5+p6+p7 + α(p4)              R3 = p0+p1+p2+p5+p6+p7+p8 + α(p1+p3+p5+p7) + α²(p4)
// (α = multiplication by X_4 on 32 planes: pure XOR; derived from
// pairwise_recursive_arithmetic.rs:18-28 applied at levels 6 and 7), transpose them back
// (the 32x32 bit transpose is an involution) and store one word column each.
// A register carries 32 elements (no [1|inf] packing); a wave-batch is 7 x 32 = 224 elements.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "bitslice.hpp"
#include "internal.hpp"

namespace bn {

namespace {
constexpr int kG = 7;            // groups per wave
constexpr int kElems = 32;       // elements per group per batch
constexpr int kWB = kG * kElems; // 224 elements per wave-batch
constexpr int kQ = 9;            // uint4 per block (32 planes + pad)
constexpr int kBlocks = 9 * kG;  // 63 partial-product blocks (>= 8*kG limb blocks)
constexpr int kZero = kBlocks;   // zero block
constexpr int kWaveQ4 = (kBlocks + 1) * kQ;
} // namespace

// Occupancy: 194 registers, two waves per SIMD.  (Until round 2 this kernel ran at one wave per SIMD with 210 values
// parked in AGPRs: the rebuild phase laundered its LDS offsets against ONE word of every four-plane result, which left
// the scheduler free to sink the other three XOR chains of every ds_read_b128 to the end of the phase -- ~200 read
// results waiting in registers.  The offsets are now laundered against all four words.)
// STRIDE (in elements, the same for a and b) is a template parameter so that row addresses are one
// base pointer plus immediates; a run-time stride makes the compiler keep 32 64-bit offsets alive
// (spilled to scratch: measured 1.3 KiB per lane).
// The lane's constants (init) and one wave-batch of the product (batch): elements e0 .. min(e0 + 224, limit) - 1 of
//   out[i] = a[i * STRIDE] * b[i * STRIDE]
template <int STRIDE>
struct mul9_wave {
	unsigned g, c, gg, w, mask, off_a[4], off_b[4], off_w, off_pp, setX, setY, setW;
	bool live, loader, builder;
	uint4 *wt;

	__device__ __forceinline__ void init(uint4 *wave_tile)
	{
	wt = wave_tile;
	const unsigned lane = threadIdx.x & 63;
	g = lane / 9;
	c = lane - g * 9;
	live = lane < 63;
	loader = live && c < 8;
	builder = live && c < 4;
	if (lane < kQ)
		wt[kZero * kQ + lane] = uint4{0, 0, 0, 0};

	w = c & 3;
	switch (c) {
	case 0: mask = 1; break;
	case 1: mask = 2; break;
	case 2: mask = 3; break;
	case 3: mask = 4; break;
	case 4: mask = 8; break;
	case 5: mask = 12; break;
	case 6: mask = 5; break;
	case 7: mask = 10; break;
	default: mask = 15; break;
	}
	if (!live) mask = 0;
#pragma unroll
	for (int s = 0; s < 4; s++) {
		const bool use = (mask >> s) & 1;
		off_a[s] = (use ? (unsigned)(s * kG + g) : (unsigned)kZero) * kQ;
		off_b[s] = (use ? (unsigned)((4 + s) * kG + g) : (unsigned)kZero) * kQ;
	}
	gg = live ? g : 0;
	off_w = (loader ? (c * kG + gg) : 0u) * kQ;
	off_pp = (live ? (c * kG + g) : (unsigned)kZero) * kQ; // where this lane publishes its partial product
	// partial products a builder needs: X (plain), Y (through alpha), W (through alpha^2); bit k = p_k
	setX = setY = setW = 0;
	if (builder) {
		switch (c) {
		case 0: setX = 0x01B; break;                              // p0 p1 p3 p4
		case 1: setX = 0x03F; setY = 0x012; break;                // p0..p5 ; alpha(p1 p4)
		case 2: setX = 0x0E3; setY = 0x010; break;                // p0 p1 p5 p6 p7 ; alpha(p4)
		default: setX = 0x1E7; setY = 0x0AA; setW = 0x010; break; // p0 p1 p2 p5 p6 p7 p8 ; alpha(p1 p3 p5 p7) ; alpha^2(p4)
		}
	}
	}

	// The slot offsets of the rebuild phase are recomputed per batch from (setX, setY, setW, g) behind
	// an opaque copy of g: hoisted out of the loop they are 15 more live registers across the
	// multiplication and push the kernel into scratch (measured: 55 us per batch instead of ~5).
	// a2 / b2 set: the operands are sums of two rows each, (a + a2) * (b + b2), added as they are loaded (the old HAL's products of
	// differences a_lo + a_hi, abi_hal.cpp round_evals_coef)
	__device__ __forceinline__ void batch(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t *__restrict__ out, uint64_t e0,
	                                      uint64_t limit, const uint32_t *__restrict__ a2 = nullptr, const uint32_t *__restrict__ b2 = nullptr)
	{
		constexpr uint64_t stride_w = (uint64_t)STRIDE << 2; // in 32-bit words
		const uint32_t *src = ((c & 4) ? b : a) + w;
		const bool full = e0 + kWB <= limit;
		{
		const uint64_t base = e0 + gg; // element of row j: base + 7*j
		uint32_t r[32];
		if (full) {
			const uint32_t *p = src + base * stride_w;
#pragma unroll
			for (int j = 0; j < 32; j++)
				r[j] = p[(uint64_t)j * 7 * stride_w];
			if (a2) { // (uniform per wave: one copy of the code serves both forms)
				const uint32_t *p2 = ((c & 4) ? b2 : a2) + w + base * stride_w;
#pragma unroll
				for (int j = 0; j < 32; j++)
					r[j] ^= p2[(uint64_t)j * 7 * stride_w];
			}
		} else {
			// ragged last batch: 8 rows at a time so only a few guarded addresses are live at once
#pragma unroll
			for (int j0 = 0; j0 < 32; j0 += 8) {
#pragma unroll
				for (int j = j0; j < j0 + 8; j++) {
					const uint64_t e = base + 7 * (uint64_t)j;
					const bool ok = e < limit;
					uint32_t v = src[ok ? e * stride_w : 0];
					if (a2) v ^= (((c & 4) ? b2 : a2) + w)[ok ? e * stride_w : 0];
					r[j] = ok ? v : 0u;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		transpose32(r);
		if (loader) {
#pragma unroll
			for (int q = 0; q < 8; q++)
				wt[off_w + q] = uint4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		uint32_t A[32], B[32];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint4 x0 = wt[off_a[0] + q], x1 = wt[off_a[1] + q], x2 = wt[off_a[2] + q], x3 = wt[off_a[3] + q];
			const uint4 y0 = wt[off_b[0] + q], y1 = wt[off_b[1] + q], y2 = wt[off_b[2] + q], y3 = wt[off_b[3] + q];
			A[4 * q] = xor3(x0.x, x1.x, x2.x) ^ x3.x;
			A[4 * q + 1] = xor3(x0.y, x1.y, x2.y) ^ x3.y;
			A[4 * q + 2] = xor3(x0.z, x1.z, x2.z) ^ x3.z;
			A[4 * q + 3] = xor3(x0.w, x1.w, x2.w) ^ x3.w;
			B[4 * q] = xor3(y0.x, y1.x, y2.x) ^ y3.x;
			B[4 * q + 1] = xor3(y0.y, y1.y, y2.y) ^ y3.y;
			B[4 * q + 2] = xor3(y0.z, y1.z, y2.z) ^ y3.z;
			B[4 * q + 3] = xor3(y0.w, y1.w, y2.w) ^ y3.w;
			if (q & 1)
				__builtin_amdgcn_sched_barrier(0);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		uint32_t P[32];
		bs_mul<5>(A, B, P);
		// publish the partial product (the limb tile is dead now: same LDS region)
#pragma unroll
		for (int q = 0; q < 8; q++)
			wt[off_pp + q] = uint4{P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]};
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// rebuild result limb c (lanes c < 4): R = X ^ alpha(Y ^ alpha(W)).  The LDS offsets are derived
		// from an opaque copy of g that is re-laundered against the previous quad's result: without that
		// chain the compiler hoists all ~100 ds_read_b128 of this phase to the top and spills their
		// results to scratch (measured: 55 us per batch).
		{
			unsigned gl = g;
			asm volatile("" : "+v"(gl));
			constexpr int ycand[5] = {1, 3, 4, 5, 7};
			uint32_t t0[32], t1[32];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const unsigned offW = (setW ? (unsigned)(4 * kG) + gl : (unsigned)kZero) * kQ;
				const uint4 ww = wt[offW + q];
				t0[4 * q] = ww.x; t0[4 * q + 1] = ww.y; t0[4 * q + 2] = ww.z; t0[4 * q + 3] = ww.w;
			}
			bs_mul_alpha<5>(t0, t1); // t1 = alpha(W)
			asm volatile("" : "+v"(gl) : "v"(t1[31]));
#pragma unroll
			for (int q = 0; q < 8; q++) {
				uint4 y{0, 0, 0, 0};
#pragma unroll
				for (int s = 0; s < 5; s++) {
					const unsigned off = (((setY >> ycand[s]) & 1) ? (unsigned)(ycand[s] * kG) + gl : (unsigned)kZero) * kQ;
					const uint4 t = wt[off + q];
					y.x ^= t.x; y.y ^= t.y; y.z ^= t.z; y.w ^= t.w;
				}
				t1[4 * q] ^= y.x; t1[4 * q + 1] ^= y.y; t1[4 * q + 2] ^= y.z; t1[4 * q + 3] ^= y.w; // Y + alpha(W)
				asm volatile("" : "+v"(gl) : "v"(t1[4 * q]), "v"(t1[4 * q + 1]), "v"(t1[4 * q + 2]), "v"(t1[4 * q + 3]));
			}
			bs_mul_alpha<5>(t1, t0); // t0 = alpha(Y) + alpha^2(W)
			asm volatile("" : "+v"(gl) : "v"(t0[31]));
#pragma unroll
			for (int q = 0; q < 8; q++) {
				uint4 x{0, 0, 0, 0};
#pragma unroll
				for (int k = 0; k < 9; k++) {
					const unsigned off = (((setX >> k) & 1) ? (unsigned)(k * kG) + gl : (unsigned)kZero) * kQ;
					const uint4 t = wt[off + q];
					x.x ^= t.x; x.y ^= t.y; x.z ^= t.z; x.w ^= t.w;
				}
				r[4 * q] = t0[4 * q] ^ x.x; r[4 * q + 1] = t0[4 * q + 1] ^ x.y; r[4 * q + 2] = t0[4 * q + 2] ^ x.z; r[4 * q + 3] = t0[4 * q + 3] ^ x.w;
				asm volatile("" : "+v"(gl) : "v"(r[4 * q]), "v"(r[4 * q + 1]), "v"(r[4 * q + 2]), "v"(r[4 * q + 3]));
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		transpose32(r); // planes -> word c of 32 elements
		if (builder) {
			uint32_t *dst = out + c + (base << 2); // builder lanes: c == word index
			if (full) {
#pragma unroll
				for (int j = 0; j < 32; j++)
					dst[28 * j] = r[j];
			} else {
#pragma unroll
				for (int j = 0; j < 32; j++)
					if (base + 7 * (uint64_t)j < limit)
						dst[28 * j] = r[j];
			}
		}
	}
}

	// ---- two wave-batches per rebuild (round 4) -------------------------------------------------------------------
	// In `batch` the rebuild of the result limbs, the transpose back and the stores are work for the four builder lanes of a
	// group -- 28 lanes of 64 -- that the whole wave executes: ~800 of the ~2450 instructions of a batch at 44 % use.  Here a
	// wave takes TWO batches (448 elements) per step: the load / transpose / limb exchange / product phases run once per
	// batch and publish their partial products into the batch's own LDS region (wt, wt + kWaveQ4), and ONE rebuild serves
	// both -- lanes c < 4 of a group rebuild limb c of the first batch, lanes 4 <= c < 8 limb c - 4 of the second --, reads
	// exactly the partial products a limb needs and folds them with three-input XORs: ~1700 instead of ~2450 instructions
	// per batch.  LDS: 18 KiB per wave, two workgroups per CU = 147 KiB (dynamic shared memory).
	// (Textually a variant of `batch` on purpose: the same statements in the same scopes.  Splitting `batch` into two
	// functions took the kernel from 192 registers to 256 + 44 spilled -- the schedule of the product is that close to the edge.)
	__device__ __forceinline__ void batch2(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t *__restrict__ out, uint64_t e0,
	                                      uint64_t limit, const uint32_t *__restrict__ a2 = nullptr, const uint32_t *__restrict__ b2 = nullptr)
	{
		constexpr uint64_t stride_w = (uint64_t)STRIDE << 2; // in 32-bit words
		uint4 *const wt0 = this->wt;
#pragma unroll 1
		for (unsigned reg = 0; reg < 2; reg++) {
		uint4 *wt = wt0 + reg * kWaveQ4;
		const uint64_t e0r = e0 + (uint64_t)reg * kWB;
		const uint32_t *src = ((c & 4) ? b : a) + w;
		const bool full = e0r + kWB <= limit;
		{
		const uint64_t base = e0r + gg; // element of row j: base + 7*j
		uint32_t r[32];
		if (full) {
			const uint32_t *p = src + base * stride_w;
#pragma unroll
			for (int j = 0; j < 32; j++)
				r[j] = p[(uint64_t)j * 7 * stride_w];
			if (a2) { // (uniform per wave: one copy of the code serves both forms)
				const uint32_t *p2 = ((c & 4) ? b2 : a2) + w + base * stride_w;
#pragma unroll
				for (int j = 0; j < 32; j++)
					r[j] ^= p2[(uint64_t)j * 7 * stride_w];
			}
		} else {
			// ragged last batch: 8 rows at a time so only a few guarded addresses are live at once
#pragma unroll
			for (int j0 = 0; j0 < 32; j0 += 8) {
#pragma unroll
				for (int j = j0; j < j0 + 8; j++) {
					const uint64_t e = base + 7 * (uint64_t)j;
					const bool ok = e < limit;
					uint32_t v = src[ok ? e * stride_w : 0];
					if (a2) v ^= (((c & 4) ? b2 : a2) + w)[ok ? e * stride_w : 0];
					r[j] = ok ? v : 0u;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		transpose32(r);
		if (loader) {
#pragma unroll
			for (int q = 0; q < 8; q++)
				wt[off_w + q] = uint4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		uint32_t A[32], B[32];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint4 x0 = wt[off_a[0] + q], x1 = wt[off_a[1] + q], x2 = wt[off_a[2] + q], x3 = wt[off_a[3] + q];
			const uint4 y0 = wt[off_b[0] + q], y1 = wt[off_b[1] + q], y2 = wt[off_b[2] + q], y3 = wt[off_b[3] + q];
			A[4 * q] = xor3(x0.x, x1.x, x2.x) ^ x3.x;
			A[4 * q + 1] = xor3(x0.y, x1.y, x2.y) ^ x3.y;
			A[4 * q + 2] = xor3(x0.z, x1.z, x2.z) ^ x3.z;
			A[4 * q + 3] = xor3(x0.w, x1.w, x2.w) ^ x3.w;
			B[4 * q] = xor3(y0.x, y1.x, y2.x) ^ y3.x;
			B[4 * q + 1] = xor3(y0.y, y1.y, y2.y) ^ y3.y;
			B[4 * q + 2] = xor3(y0.z, y1.z, y2.z) ^ y3.z;
			B[4 * q + 3] = xor3(y0.w, y1.w, y2.w) ^ y3.w;
			if (q & 1)
				__builtin_amdgcn_sched_barrier(0);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		uint32_t P[32];
		bs_mul<5>(A, B, P);
		// publish the partial product (the limb tile is dead now: same LDS region)
#pragma unroll
		for (int q = 0; q < 8; q++)
			wt[off_pp + q] = uint4{P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]};
		}
		} // (both batches' partial products are in LDS)
		{
		uint32_t r[32];
		uint4 *wt = wt0;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// rebuild: lane c < 8 of a group builds limb c & 3 of batch c >> 2 -- R = X ^ alpha(Y ^ alpha(W)) with exactly the partial
		// products the limb needs (7 + 4 + 1 blocks per four planes, padded with the zero block).  The lists are packed four bits
		// per entry (9 = zero block) and unpacked into twelve offsets HERE, behind the product, from an opaque copy of g.
		{
			unsigned gl = g;
			asm volatile("" : "+v"(gl));
			uint32_t xl, yl, wl;
			switch ((live && c < 8) ? (c & 3) : 4u) {
			case 0: xl = 0x9994310u; yl = 0x9999u; wl = 9; break; // p0 p1 p3 p4
			case 1: xl = 0x9543210u; yl = 0x9941u; wl = 9; break; // p0..p5 ; alpha(p1 p4)
			case 2: xl = 0x9976510u; yl = 0x9994u; wl = 9; break; // p0 p1 p5 p6 p7 ; alpha(p4)
			case 3: xl = 0x8765210u; yl = 0x7531u; wl = 4; break; // p0 p1 p2 p5 p6 p7 p8 ; alpha(p1 p3 p5 p7) ; alpha^2(p4)
			default: xl = 0x9999999u; yl = 0x9999u; wl = 9; break;
			}
			const unsigned reg_q = (c & 4) ? (unsigned)kWaveQ4 : 0u;
			auto off_of = [&](uint32_t idx) -> unsigned { return (idx == 9 ? (unsigned)kZero : idx * kG + gl) * kQ + reg_q; };
			unsigned ox[7], oy[4], ow;
#pragma unroll
			for (int k = 0; k < 7; k++) ox[k] = off_of((xl >> (4 * k)) & 15u);
#pragma unroll
			for (int k = 0; k < 4; k++) oy[k] = off_of((yl >> (4 * k)) & 15u);
			ow = off_of(wl);
			uint32_t t0[32], t1[32];
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const uint4 ww = wt[ow + q];
				t0[4 * q] = ww.x; t0[4 * q + 1] = ww.y; t0[4 * q + 2] = ww.z; t0[4 * q + 3] = ww.w;
			}
			bs_mul_alpha<5>(t0, t1); // t1 = alpha(W)
			asm volatile("" : "+v"(oy[0]), "+v"(oy[1]), "+v"(oy[2]), "+v"(oy[3]) : "v"(t1[31]));
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const uint4 y0 = wt[oy[0] + q], y1 = wt[oy[1] + q], y2 = wt[oy[2] + q], y3 = wt[oy[3] + q];
				t1[4 * q] = xor3(xor3(y0.x, y1.x, y2.x), y3.x, t1[4 * q]);
				t1[4 * q + 1] = xor3(xor3(y0.y, y1.y, y2.y), y3.y, t1[4 * q + 1]);
				t1[4 * q + 2] = xor3(xor3(y0.z, y1.z, y2.z), y3.z, t1[4 * q + 2]);
				t1[4 * q + 3] = xor3(xor3(y0.w, y1.w, y2.w), y3.w, t1[4 * q + 3]); // Y + alpha(W)
				// (the next quad's reads wait for this quad's result: left alone the compiler hoists every read of the phase)
				asm volatile("" : "+v"(oy[0]), "+v"(oy[1]), "+v"(oy[2]), "+v"(oy[3]) : "v"(t1[4 * q]), "v"(t1[4 * q + 1]), "v"(t1[4 * q + 2]), "v"(t1[4 * q + 3]));
			}
			bs_mul_alpha<5>(t1, t0); // t0 = alpha(Y) + alpha^2(W)
			asm volatile("" : "+v"(ox[0]), "+v"(ox[1]), "+v"(ox[2]), "+v"(ox[3]), "+v"(ox[4]), "+v"(ox[5]), "+v"(ox[6]) : "v"(t0[31]));
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const uint4 x0 = wt[ox[0] + q], x1 = wt[ox[1] + q], x2 = wt[ox[2] + q], x3 = wt[ox[3] + q];
				const uint4 x4 = wt[ox[4] + q], x5 = wt[ox[5] + q], x6 = wt[ox[6] + q];
				r[4 * q] = xor3(xor3(x0.x, x1.x, x2.x), xor3(x3.x, x4.x, x5.x), x6.x ^ t0[4 * q]);
				r[4 * q + 1] = xor3(xor3(x0.y, x1.y, x2.y), xor3(x3.y, x4.y, x5.y), x6.y ^ t0[4 * q + 1]);
				r[4 * q + 2] = xor3(xor3(x0.z, x1.z, x2.z), xor3(x3.z, x4.z, x5.z), x6.z ^ t0[4 * q + 2]);
				r[4 * q + 3] = xor3(xor3(x0.w, x1.w, x2.w), xor3(x3.w, x4.w, x5.w), x6.w ^ t0[4 * q + 3]);
				asm volatile("" : "+v"(ox[0]), "+v"(ox[1]), "+v"(ox[2]), "+v"(ox[3]), "+v"(ox[4]), "+v"(ox[5]), "+v"(ox[6])
				             : "v"(r[4 * q]), "v"(r[4 * q + 1]), "v"(r[4 * q + 2]), "v"(r[4 * q + 3]));
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		transpose32(r); // planes -> word c & 3 of the 32 elements of this group in batch c >> 2
		if (live && c < 8) {
			const uint64_t base = e0 + (uint64_t)(c >> 2) * kWB + gg;
			uint32_t *dst = out + (c & 3) + (base << 2);
			if (e0 + 2 * kWB <= limit) {
#pragma unroll
				for (int j = 0; j < 32; j++)
					dst[28 * j] = r[j];
			} else {
#pragma unroll
				for (int j = 0; j < 32; j++)
					if (base + 7 * (uint64_t)j < limit)
						dst[28 * j] = r[j];
			}
		}
		}
	}

};

// The wave-batches wave_global, wave_global + n_waves, ... of one element-wise product; wt = this wave's LDS tile.
template <int STRIDE>
__device__ __forceinline__ void mul9_batches(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, uint32_t *__restrict__ out,
                                             uint64_t n, uint64_t wave_global, uint64_t n_waves, uint4 *wt)
{
	mul9_wave<STRIDE> mw;
	mw.init(wt);
	const uint64_t n_batches = (n + kWB - 1) / kWB;
	for (uint64_t bt = wave_global; bt < n_batches; bt += n_waves)
		mw.batch(a, b, out, bt * kWB, n);
}

template <int STRIDE>
__global__ __launch_bounds__(256, 2) void k_mul9_dual(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                     uint32_t *__restrict__ out, uint64_t n)
{
	extern __shared__ uint4 tile2[];
	const unsigned wave = threadIdx.x >> 6;
	mul9_wave<STRIDE> mw;
	mw.init(tile2 + wave * 2 * kWaveQ4);
	if ((threadIdx.x & 63) < kQ)
		tile2[wave * 2 * kWaveQ4 + kWaveQ4 + kZero * kQ + (threadIdx.x & 63)] = uint4{0, 0, 0, 0}; // the second region's zero block
	const uint64_t n_steps = (n + 2 * kWB - 1) / (2 * kWB);
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;
	for (uint64_t st = (uint64_t)blockIdx.x * 4 + wave; st < n_steps; st += n_waves)
		mw.batch2(a, b, out, st * 2 * kWB, n);
}

template <int STRIDE>
__global__ __launch_bounds__(256, 2) void k_mul9(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                uint32_t *__restrict__ out, uint64_t n)
{
	__shared__ uint4 tile[4][kWaveQ4];
	const unsigned wave = threadIdx.x >> 6;
	mul9_batches<STRIDE>(a, b, out, n, (uint64_t)blockIdx.x * 4 + wave, (uint64_t)gridDim.x * 4, tile[wave]);
}

// Several levels of pairwise_product_reduce in one launch: a workgroup takes 4 adjacent wave-batches of the first level
// (896 products), and -- their results being exactly the inputs of the 448 products under them -- goes on with 2 batches of
// the next level, 1 of the third, half a batch of the fourth, before it moves to its next 896.  What a level reads was
// stored by waves of the same workgroup (same CU, same L1: a workgroup barrier orders it); no launch between the levels
// and no grid-wide dependency.  Level l (0-based) reads lv[l - 1] (level 0: in) and writes lv[l]; n0 = products of level 0.
__global__ __launch_bounds__(256, 2) void k_mul9_tree(mul9_tree_args args)
{
	__shared__ uint4 tile[4][kWaveQ4];
	const unsigned wave = threadIdx.x >> 6;
	mul9_wave<2> mw;
	mw.init(tile[wave]);
	constexpr uint64_t kSB = 4 * kWB; // products of the first level per workgroup step
	const uint64_t n_sb = (args.n0 + kSB - 1) / kSB;
	for (uint64_t k = blockIdx.x; k < n_sb; k += gridDim.x) {
		for (uint32_t l = 0; l < args.n_levels; l++) {
			if (l) __syncthreads();
			const uint64_t span = kSB >> l;           // products of level l under this step
			const uint64_t n_l = args.n0 >> l;        // products of level l in all
			const uint64_t e0 = span * k + (uint64_t)kWB * wave;
			uint64_t limit = span * (k + 1);
			if (limit > n_l) limit = n_l;
			if (e0 < limit) {
				const uint32_t *src = l ? args.lv[l - 1] : args.in;
				mw.batch(src, src + 4, args.lv[l], e0, limit);
			}
		}
		__syncthreads(); // (the LDS tiles are per wave, but the next step's first level must not overtake a wave still reading)
	}
}

// n_jobs products of n elements each in one launch: wave-batches are dealt out over (job, batch) pairs; a wave's job pointers are
// read from the table (pinned memory) and held in scalar registers
__global__ __launch_bounds__(256, 2) void k_mul9_jobs(const mul9_job *__restrict__ jobs, uint32_t n_jobs, uint64_t n)
{
	__shared__ uint4 tile[4][kWaveQ4];
	const unsigned wave = threadIdx.x >> 6;
	mul9_wave<1> mw;
	mw.init(tile[wave]);
	const uint64_t per_job = (n + kWB - 1) / kWB, total = per_job * n_jobs;
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;
	// (the table is pinned host memory: a wave reads a job's pointers once, when it moves on to that job)
	uint32_t cur = ~0u;
	const uint32_t *a = nullptr, *b = nullptr, *a2 = nullptr, *b2 = nullptr;
	uint32_t *out = nullptr;
	for (uint64_t bt = (uint64_t)blockIdx.x * 4 + wave; bt < total; bt += n_waves) {
		const uint32_t j = (uint32_t)(bt / per_job);
		const uint64_t e0 = (bt - (uint64_t)j * per_job) * kWB;
		auto uni = [](const void *p) { // (uniform per wave: into scalar registers)
			const uint64_t v = (uint64_t)p;
			return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
		};
		if (j != cur) {
			cur = j;
			a = (const uint32_t *)uni(jobs[j].a);
			b = (const uint32_t *)uni(jobs[j].b);
			out = (uint32_t *)uni(jobs[j].out);
			a2 = (const uint32_t *)uni(jobs[j].a2);
			b2 = (const uint32_t *)uni(jobs[j].b2);
		}
		mw.batch(a, b, out, e0, n, a2, b2);
	}
}

// the same with two wave-batches per rebuild (batch2: a fifth fewer instructions per product), for launches with more batches than
// wave slots
__global__ __launch_bounds__(256, 2) void k_mul9_jobs_dual(const mul9_job *__restrict__ jobs, uint32_t n_jobs, uint64_t n)
{
	extern __shared__ uint4 tile2[];
	const unsigned wave = threadIdx.x >> 6;
	mul9_wave<1> mw;
	mw.init(tile2 + wave * 2 * kWaveQ4);
	if ((threadIdx.x & 63) < kQ)
		tile2[wave * 2 * kWaveQ4 + kWaveQ4 + kZero * kQ + (threadIdx.x & 63)] = uint4{0, 0, 0, 0}; // the second region's zero block
	const uint64_t per_job = (n + 2 * kWB - 1) / (2 * kWB), total = per_job * n_jobs;
	const uint64_t n_waves = (uint64_t)gridDim.x * 4;
	uint32_t cur = ~0u;
	const uint32_t *a = nullptr, *b = nullptr, *a2 = nullptr, *b2 = nullptr;
	uint32_t *out = nullptr;
	for (uint64_t st = (uint64_t)blockIdx.x * 4 + wave; st < total; st += n_waves) {
		const uint32_t j = (uint32_t)(st / per_job);
		const uint64_t e0 = (st - (uint64_t)j * per_job) * 2 * kWB;
		auto uni = [](const void *p) {
			const uint64_t v = (uint64_t)p;
			return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
		};
		if (j != cur) {
			cur = j;
			a = (const uint32_t *)uni(jobs[j].a);
			b = (const uint32_t *)uni(jobs[j].b);
			out = (uint32_t *)uni(jobs[j].out);
			a2 = (const uint32_t *)uni(jobs[j].a2);
			b2 = (const uint32_t *)uni(jobs[j].b2);
		}
		mw.batch2(a, b, out, e0, n, a2, b2);
	}
}

hipError_t launch_mul9_jobs(hipStream_t s, int n_cu, const mul9_job *d_jobs, uint32_t n_jobs, uint64_t n)
{
	if (n_jobs == 0 || n == 0) return hipSuccess;
	const uint64_t total = ((n + kWB - 1) / kWB) * n_jobs;
	uint64_t blocks = (total + 3) / 4;
	const uint64_t cap = (uint64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	__atomic_thread_fence(__ATOMIC_SEQ_CST); // (the table is in memory before the doorbell rings)
	if (total > cap * 4) {
		constexpr size_t lds = (size_t)4 * 2 * kWaveQ4 * sizeof(uint4);
		const hipError_t attr1 = func_lds_limit(reinterpret_cast<const void *>(&k_mul9_jobs_dual), (int)lds);
		if (attr1 != hipSuccess) return attr1;
		const uint64_t steps = ((n + 2 * kWB - 1) / (2 * kWB)) * n_jobs;
		uint64_t blk = (steps + 3) / 4;
		if (blk > cap) blk = cap;
		hipLaunchKernelGGL(k_mul9_jobs_dual, dim3((unsigned)blk), dim3(256), lds, s, d_jobs, n_jobs, n);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_mul9_jobs, dim3((unsigned)blocks), dim3(256), 0, s, d_jobs, n_jobs, n);
	return hipGetLastError();
}

hipError_t launch_mul9(hipStream_t s, int n_cu, const void *a, uint64_t a_stride, const void *b, uint64_t b_stride, uint64_t b_off,
                       void *out, uint64_t n)
{
	if (n == 0) return hipSuccess;
	const uint64_t n_batches = (n + kWB - 1) / kWB;
	uint64_t blocks = (n_batches + 3) / 4;
	const uint64_t cap = (uint64_t)n_cu * 2; // two workgroups per CU = two waves per SIMD
	if (blocks > cap) blocks = cap;
	const uint32_t *pb = (const uint32_t *)b + b_off * 4;
	// more batches than wave slots: two batches per rebuild (k_mul9_dual); BN_MUL9_DUAL=0 keeps the one-batch kernel
	static const bool dual_on = [] {
		const char *e = bn::settled_knob("BN_MUL9_DUAL");
		return !(e && e[0] == '0');
	}();
	// (unit strides only: the strided form -- one level of pairwise_product_reduce by itself -- never has that many batches
	// with the default level fusion, and a path the tests do not reach is not worth a fifth of its time)
	if (dual_on && n_batches > cap * 4 && a_stride == 1 && b_stride == 1) {
		constexpr size_t lds = (size_t)4 * 2 * kWaveQ4 * sizeof(uint4);
		const hipError_t attr1 = func_lds_limit(reinterpret_cast<const void *>(&k_mul9_dual<1>), (int)lds);
		if (attr1 != hipSuccess) return attr1;
		const uint64_t n_steps = (n_batches + 1) / 2;
		uint64_t blk = (n_steps + 3) / 4;
		if (blk > cap) blk = cap;
		hipLaunchKernelGGL(k_mul9_dual<1>, dim3((unsigned)blk), dim3(256), lds, s, (const uint32_t *)a, pb, (uint32_t *)out, n);
		return hipGetLastError();
	}
	if (a_stride == 1 && b_stride == 1)
		hipLaunchKernelGGL(k_mul9<1>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t *)a, pb, (uint32_t *)out, n);
	else if (a_stride == 2 && b_stride == 2)
		hipLaunchKernelGGL(k_mul9<2>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint32_t *)a, pb, (uint32_t *)out, n);
	else
		return hipErrorInvalidValue;
	return hipGetLastError();
}

hipError_t launch_mul9_tree(hipStream_t s, int n_cu, const mul9_tree_args &args)
{
	if (args.n0 == 0 || args.n_levels == 0 || args.n_levels > 4) return hipErrorInvalidValue;
	uint64_t blocks = (args.n0 + 4 * kWB - 1) / (4 * kWB);
	const uint64_t cap = (uint64_t)n_cu * 2;
	if (blocks > cap) blocks = cap;
	hipLaunchKernelGGL(k_mul9_tree, dim3((unsigned)blocks), dim3(256), 0, s, args);
	return hipGetLastError();
}

} // namespace bn
