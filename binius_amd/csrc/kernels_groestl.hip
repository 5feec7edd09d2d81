// binius_amd/csrc/kernels_groestl.hip -- Groestl-256 leaf hashing, the Groestl 2-to-1 compression and
// the binary Merkle tree of the reference's vector commitment, on device.
//
// What it replaces: BinaryMerkleTreeProver::commit (crates/core/src/merkle_tree/prover.rs:47-62) ->
// binary_merkle_tree::build (binary_merkle_tree.rs:27-101) with H = Groestl256
// (crates/hash/src/groestl/digest.rs:30-87) and C = Groestl256ByteCompression (compression.rs:21-36),
// which the FRI prover calls on every folded codeword after copying it to the host
// (crates/core/src/protocols/fri/prove.rs:395-420).  With the tree built on device only the
// 64 B-per-leaf node array crosses PCIe for the commitment, not the codeword.
//
// Shape.  One lane = one hash instance: its state is the 8 x 8 byte matrix as eight little-endian
// 64-bit columns (16 VGPRs); a round is AddRoundConstant on the columns, then SubBytes + ShiftBytes +
// MixBytes as 64 lookups of the fused column table T0[b] = S(b) * (02 02 03 04 05 03 05 07)^T
// (the other seven row tables are byte rotations of T0: v_alignbyte).  Byte work, no GEMM shape; the
// bound is LDS lookups + VALU issue, far under the HBM roofline (~150 lane-ops per message byte).
//
// LDS.  T0 is 2 KiB; it is replicated 32 times (64 KiB), copy c = lane & 31 at byte offset
// b * 256 + c * 8, so that every half-wave ds_read_b64 touches 32 distinct bank pairs whatever the
// data is: no bank conflicts (a single shared copy serialises ~3.5x on random bytes).  The lookup
// address {0, 0, byte, lane offset} is one v_perm_b32.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "internal.hpp"

namespace bn {

namespace {

// ---- the column table, generated at compile time from the definitions (specification 3.4.3, 3.4.5)
struct groestl_t0 {
	uint64_t v[256];
};

constexpr uint8_t gr_xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x & 0x80) ? 0x1B : 0)); }
constexpr uint8_t gr_rotl8(uint8_t x, int s) { return (uint8_t)((x << s) | (x >> (8 - s))); }

constexpr groestl_t0 make_groestl_t0()
{
	uint8_t sbox[256] = {};
	uint8_t p = 1, q = 1;
	do { // p runs over the multiplicative group of GF(2^8) (times 3), q = 1 / p
		p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1B : 0));
		q = (uint8_t)(q ^ (q << 1));
		q = (uint8_t)(q ^ (q << 2));
		q = (uint8_t)(q ^ (q << 4));
		if (q & 0x80) q = (uint8_t)(q ^ 0x09);
		sbox[p] = (uint8_t)(q ^ gr_rotl8(q, 1) ^ gr_rotl8(q, 2) ^ gr_rotl8(q, 3) ^ gr_rotl8(q, 4) ^ 0x63);
	} while (p != 1);
	sbox[0] = 0x63;
	// byte r of T0[b] = circ[(0 - r) mod 8] * S(b): what a byte in row 0 of a column adds to row r
	const int circ[8] = {2, 2, 3, 4, 5, 3, 5, 7};
	groestl_t0 t{};
	for (int b = 0; b < 256; b++) {
		const uint8_t s1 = sbox[b], s2 = gr_xtime(s1), s4 = gr_xtime(s2);
		uint64_t w = 0;
		for (int r = 0; r < 8; r++) {
			const int m = circ[(8 - r) & 7];
			const uint8_t e = (uint8_t)(((m & 1) ? s1 : 0) ^ ((m & 2) ? s2 : 0) ^ ((m & 4) ? s4 : 0));
			w |= (uint64_t)e << (8 * r);
		}
		t.v[b] = w;
	}
	return t;
}

__constant__ groestl_t0 kGroestlT0 = make_groestl_t0();

constexpr int kCopies = 32;
constexpr unsigned kTableBytes = 256 * kCopies * 8; // 64 KiB

__device__ __forceinline__ void stage_table(uint2 *tab)
{
	for (unsigned i = threadIdx.x; i < 256u * kCopies; i += blockDim.x) {
		const uint64_t w = kGroestlT0.v[i >> 5];
		tab[i] = uint2{(uint32_t)w, (uint32_t)(w >> 32)};
	}
	__syncthreads();
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// the state: column c = (lo[c], hi[c]), row r of the column = byte r of the 64-bit value
struct gstate {
	uint32_t lo[8], hi[8];
};

// The table sits at LDS address 0 (these kernels have no static __shared__; checked at kernel entry),
// so the lookup address IS the LDS address: reading through an integer-made LDS pointer keeps the
// compiler from adding the (zero) dynamic-LDS base to every address -- one VALU less per lookup.
typedef unsigned int gr_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const gr_u2 lds_cuint2;

__device__ __forceinline__ void require_table_at_lds_zero(const void *smem)
{
	if ((uint32_t)(size_t)(__attribute__((address_space(3))) const void *)smem != 0u) __builtin_trap();
}

// table value of byte K (0..7) of the column word pair (lo, hi)
template <int K>
__device__ __forceinline__ uint2 lookup(const char *, uint32_t lo, uint32_t hi, uint32_t lane_off)
{
	// address = byte << 8 | lane_off: bytes {lane_off.b0, word.bK, 0, 0}
	const uint32_t word = K < 4 ? lo : hi;
	constexpr uint32_t sel = 0x0C0C0000u | ((4u + (K & 3)) << 8) | 0u;
	const uint32_t addr = __builtin_amdgcn_perm(word, lane_off, sel);
	const gr_u2 v = *(lds_cuint2 *)(size_t)addr;
	return uint2{v.x, v.y};
}

// ROTL64(t, 8K)
template <int K>
__device__ __forceinline__ uint2 rot_bytes(uint2 t)
{
	if constexpr (K == 0) return t;
	else if constexpr (K == 4) return uint2{t.y, t.x};
	else if constexpr (K < 4)
		return uint2{__builtin_amdgcn_alignbyte(t.x, t.y, 4 - K), __builtin_amdgcn_alignbyte(t.y, t.x, 4 - K)};
	else
		return uint2{__builtin_amdgcn_alignbyte(t.y, t.x, 8 - K), __builtin_amdgcn_alignbyte(t.x, t.y, 8 - K)};
}

// XOR_k ROTL64(t_k, 8k): rows k and k + 4 differ by a half swap, which is free, so they are paired
// BEFORE the byte rotation -- three rotations per column instead of six
__device__ __forceinline__ uint2 mix_rows(uint2 t0, uint2 t1, uint2 t2, uint2 t3, uint2 t4, uint2 t5, uint2 t6, uint2 t7)
{
	const uint2 r1 = rot_bytes<1>(uint2{t1.x ^ t5.y, t1.y ^ t5.x});
	const uint2 r2 = rot_bytes<2>(uint2{t2.x ^ t6.y, t2.y ^ t6.x});
	const uint2 r3 = rot_bytes<3>(uint2{t3.x ^ t7.y, t3.y ^ t7.x});
	return uint2{xor3(xor3(t0.x, t4.y, r1.x), r2.x, r3.x), xor3(xor3(t0.y, t4.x, r1.y), r2.y, r3.y)};
}

template <bool Q>
struct shifts;
template <>
struct shifts<false> {
	static constexpr int s[8] = {0, 1, 2, 3, 4, 5, 6, 7};
};
template <>
struct shifts<true> {
	static constexpr int s[8] = {1, 3, 5, 7, 0, 2, 4, 6};
};

// one output column: XOR_k T_k[row k of column (C + sigma_k) mod 8]
template <bool Q, int C>
__device__ __forceinline__ void column(const gstate &in, gstate &out, const char *tab, uint32_t lane_off)
{
	using S = shifts<Q>;
	const uint2 t0 = lookup<0>(tab, in.lo[(C + S::s[0]) & 7], in.hi[(C + S::s[0]) & 7], lane_off);
	const uint2 t1 = lookup<1>(tab, in.lo[(C + S::s[1]) & 7], in.hi[(C + S::s[1]) & 7], lane_off);
	const uint2 t2 = lookup<2>(tab, in.lo[(C + S::s[2]) & 7], in.hi[(C + S::s[2]) & 7], lane_off);
	const uint2 t3 = lookup<3>(tab, in.lo[(C + S::s[3]) & 7], in.hi[(C + S::s[3]) & 7], lane_off);
	const uint2 t4 = lookup<4>(tab, in.lo[(C + S::s[4]) & 7], in.hi[(C + S::s[4]) & 7], lane_off);
	const uint2 t5 = lookup<5>(tab, in.lo[(C + S::s[5]) & 7], in.hi[(C + S::s[5]) & 7], lane_off);
	const uint2 t6 = lookup<6>(tab, in.lo[(C + S::s[6]) & 7], in.hi[(C + S::s[6]) & 7], lane_off);
	const uint2 t7 = lookup<7>(tab, in.lo[(C + S::s[7]) & 7], in.hi[(C + S::s[7]) & 7], lane_off);
	const uint2 o = mix_rows(t0, t1, t2, t3, t4, t5, t6, t7);
	out.lo[C] = o.x;
	out.hi[C] = o.y;
}

template <bool Q>
__device__ __forceinline__ void round_fn(gstate &s, uint32_t rnd, const char *tab, uint32_t lane_off)
{
	// AddRoundConstant (3.4.2): P: row 0 of column c ^= (c << 4) ^ round;
	// Q: everything ^= 0xFF, row 7 of column c additionally ^= (c << 4) ^ round
#pragma unroll
	for (int c = 0; c < 8; c++) {
		if constexpr (!Q) {
			s.lo[c] ^= (uint32_t)(c << 4) ^ rnd;
		} else {
			s.lo[c] = ~s.lo[c];
			s.hi[c] = ~s.hi[c] ^ (((uint32_t)(c << 4) ^ rnd) << 24);
		}
	}
	gstate n;
	column<Q, 0>(s, n, tab, lane_off);
	column<Q, 1>(s, n, tab, lane_off);
	column<Q, 2>(s, n, tab, lane_off);
	column<Q, 3>(s, n, tab, lane_off);
	column<Q, 4>(s, n, tab, lane_off);
	column<Q, 5>(s, n, tab, lane_off);
	column<Q, 6>(s, n, tab, lane_off);
	column<Q, 7>(s, n, tab, lane_off);
	s = n;
}

// P alone (output transformation, 2-to-1 compression)
__device__ __forceinline__ void perm_p(gstate &s, const char *tab, uint32_t lane_off)
{
#pragma unroll 1
	for (uint32_t r = 0; r < 10; r++) round_fn<false>(s, r, tab, lane_off);
}

// Q alone: the padding block of a leaf whose length is a multiple of 64 bytes is the same for every leaf of a launch -- Q(m_pad) once
// per lane instead of once per leaf
__device__ __forceinline__ void perm_q(gstate &s, const char *tab, uint32_t lane_off)
{
#pragma unroll 1
	for (uint32_t r = 0; r < 10; r++) round_fn<true>(s, r, tab, lane_off);
}

// h <- P(h ^ m) ^ q ^ h with q = Q(m) given
__device__ __forceinline__ void compress_known_q(gstate &h, const gstate &m, const gstate &q, const char *tab, uint32_t lane_off)
{
	gstate p;
#pragma unroll
	for (int c = 0; c < 8; c++) {
		p.lo[c] = h.lo[c] ^ m.lo[c];
		p.hi[c] = h.hi[c] ^ m.hi[c];
	}
	perm_p(p, tab, lane_off);
#pragma unroll
	for (int c = 0; c < 8; c++) {
		h.lo[c] ^= p.lo[c] ^ q.lo[c];
		h.hi[c] ^= p.hi[c] ^ q.hi[c];
	}
}

// h <- P(h ^ m) ^ Q(m) ^ h (crates/hash/src/groestl/mod.rs:26-34); the two permutations advance together:
// two independent dependency chains per lane
__device__ __forceinline__ void compress(gstate &h, const gstate &m, const char *tab, uint32_t lane_off)
{
	gstate p, q;
#pragma unroll
	for (int c = 0; c < 8; c++) {
		p.lo[c] = h.lo[c] ^ m.lo[c];
		p.hi[c] = h.hi[c] ^ m.hi[c];
		q.lo[c] = m.lo[c];
		q.hi[c] = m.hi[c];
	}
#pragma unroll 1
	for (uint32_t r = 0; r < 10; r++) {
		round_fn<false>(p, r, tab, lane_off);
		round_fn<true>(q, r, tab, lane_off);
	}
#pragma unroll
	for (int c = 0; c < 8; c++) {
		h.lo[c] ^= p.lo[c] ^ q.lo[c];
		h.hi[c] ^= p.hi[c] ^ q.hi[c];
	}
}

// Omega(x) = the last 32 bytes of P(x) ^ x = columns 4..7
__device__ __forceinline__ void output_transform(const gstate &x, uint4 &d0, uint4 &d1, const char *tab, uint32_t lane_off)
{
	gstate p = x;
	perm_p(p, tab, lane_off);
	d0 = uint4{p.lo[4] ^ x.lo[4], p.hi[4] ^ x.hi[4], p.lo[5] ^ x.lo[5], p.hi[5] ^ x.hi[5]};
	d1 = uint4{p.lo[6] ^ x.lo[6], p.hi[6] ^ x.hi[6], p.lo[7] ^ x.lo[7], p.hi[7] ^ x.hi[7]};
}

__device__ __forceinline__ void set_cols(gstate &m, int c0, uint4 v)
{
	m.lo[c0] = v.x;
	m.hi[c0] = v.y;
	m.lo[c0 + 1] = v.z;
	m.hi[c0 + 1] = v.w;
}

constexpr int kThreads = 512;

// digest[leaf] = Groestl-256(elems[leaf * batch .. (leaf + 1) * batch) as 16 * batch bytes)
// (binary_merkle_tree.rs:175-211 hash_interleaved; digest.rs:62-87 padding: 16 * batch mod 64 is at
// most 48 < 56, so there is always exactly one padding block and the block count is full + 1)
__global__ __launch_bounds__(kThreads) void k_groestl_leaves(const uint4 *__restrict__ elems, uint64_t batch, uint64_t n_leaves,
                                                             uint4 *__restrict__ digests)
{
	extern __shared__ __align__(16) unsigned char smem[];
	require_table_at_lds_zero(smem);
	stage_table(reinterpret_cast<uint2 *>(smem));
	const char *tab = reinterpret_cast<const char *>(smem);
	const uint32_t lane_off = (threadIdx.x & (kCopies - 1)) * 8;
	const uint64_t n_full = batch >> 2; // 64-byte blocks
	const uint32_t rem = (uint32_t)(batch & 3);
	uint32_t cnt = (uint32_t)(n_full + 1); // (a leaf of 2^38 bytes does not exist: 32 bits are enough)
	cnt = __builtin_bswap32(cnt);
	// leaves of whole blocks (rem == 0: the interleaved codewords' 2^k elements): the padding block is a launch constant, and so is
	// Q of it -- ten permutations per leaf of four blocks instead of eleven
	gstate m_pad, q_pad;
#pragma unroll
	for (int c = 0; c < 8; c++) m_pad.lo[c] = m_pad.hi[c] = q_pad.lo[c] = q_pad.hi[c] = 0;
	const bool whole = rem == 0;
	if (whole && (uint64_t)blockIdx.x * kThreads + threadIdx.x < n_leaves) {
		m_pad.lo[0] = 0x80;
		m_pad.hi[7] = cnt;
		q_pad = m_pad;
		perm_q(q_pad, tab, lane_off);
	}
	for (uint64_t leaf = (uint64_t)blockIdx.x * kThreads + threadIdx.x; leaf < n_leaves; leaf += (uint64_t)gridDim.x * kThreads) {
		const uint4 *src = elems + leaf * batch;
		gstate h;
#pragma unroll
		for (int c = 0; c < 8; c++) h.lo[c] = h.hi[c] = 0;
		h.hi[7] = 0x00010000u; // byte 62 = 0x01: the output length 256 as a big-endian 64-bit integer in column 7
		for (uint64_t b = 0; b < n_full; b++) {
			gstate m;
			set_cols(m, 0, src[4 * b]);
			set_cols(m, 2, src[4 * b + 1]);
			set_cols(m, 4, src[4 * b + 2]);
			set_cols(m, 6, src[4 * b + 3]);
			compress(h, m, tab, lane_off);
		}
		if (whole) {
			compress_known_q(h, m_pad, q_pad, tab, lane_off);
		} else {
			gstate m;
#pragma unroll
			for (int c = 0; c < 8; c++) m.lo[c] = m.hi[c] = 0;
			const uint4 *tail = src + 4 * n_full;
			if (rem > 0) set_cols(m, 0, tail[0]);
			if (rem > 1) set_cols(m, 2, tail[1]);
			if (rem > 2) set_cols(m, 4, tail[2]);
			// 0x80 right after the message: byte 16 * rem = row 0 of column 2 * rem
			if (rem == 0) m.lo[0] = 0x80;
			if (rem == 1) m.lo[2] = 0x80;
			if (rem == 2) m.lo[4] = 0x80;
			if (rem == 3) m.lo[6] = 0x80;
			m.hi[7] = cnt; // the block count, big-endian, in the last 8 bytes
			compress(h, m, tab, lane_off);
		}
		uint4 d0, d1;
		output_transform(h, d0, d1, tab, lane_off);
		digests[2 * leaf] = d0;
		digests[2 * leaf + 1] = d1;
	}
}

// next[i] = last 32 bytes of P(x) ^ x, x = prev[2i] || prev[2i+1]  (compression.rs:21-36,
// binary_merkle_tree.rs:158-168 compress_layer)
__global__ __launch_bounds__(kThreads) void k_groestl_layer(const uint4 *__restrict__ prev, uint64_t n_out, uint4 *__restrict__ next)
{
	extern __shared__ __align__(16) unsigned char smem[];
	require_table_at_lds_zero(smem);
	stage_table(reinterpret_cast<uint2 *>(smem));
	const char *tab = reinterpret_cast<const char *>(smem);
	const uint32_t lane_off = (threadIdx.x & (kCopies - 1)) * 8;
	for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n_out; i += (uint64_t)gridDim.x * kThreads) {
		gstate x;
		set_cols(x, 0, prev[4 * i]);
		set_cols(x, 2, prev[4 * i + 1]);
		set_cols(x, 4, prev[4 * i + 2]);
		set_cols(x, 6, prev[4 * i + 3]);
		uint4 d0, d1;
		output_transform(x, d0, d1, tab, lane_off);
		next[2 * i] = d0;
		next[2 * i + 1] = d1;
	}
}

// ---- the top of the tree in one workgroup, eight lanes per hash.
// The upper levels are a chain of dependent permutations with little parallelism: what counts is the
// latency of one P, not throughput.  One lane per hash spends ~13 us per level (64 dependent lookups and
// ~250 VALU per round in ONE wave); here the eight columns of a state sit in eight lanes, a round is
// 8 lookups + ~36 VALU per lane, and the ShiftBytes traffic between columns is DPP: two hashes share a
// 16-lane row, hash j = lane & 1, column c = (lane & 15) >> 1, so "the column sigma_k = k further on"
// is row_ror by 16 - 2k.  Levels are exchanged through two LDS buffers; every level is also written to
// its place behind `layer` (the flattened order of binary_merkle_tree.rs:22-25).
template <int SIGMA>
__device__ __forceinline__ uint32_t from_column_plus(uint32_t v)
{
	if constexpr (SIGMA == 0) return v;
	else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + (16 - 2 * SIGMA), 0xF, 0xF, false); // row_ror:16-2*sigma
}

template <bool Q, int K>
__device__ __forceinline__ uint2 lane_lookup(const char *tab, uint32_t lo, uint32_t hi, uint32_t lane_off)
{
	const uint32_t w = from_column_plus<shifts<Q>::s[K]>(K < 4 ? lo : hi);
	return lookup<K>(tab, w, w, lane_off);
}

// one round of P or Q on a state spread over eight lanes: (lo, hi) = this lane's column c
template <bool Q>
__device__ __forceinline__ void round_lanes(uint32_t &lo, uint32_t &hi, uint32_t c, uint32_t r, const char *tab, uint32_t lane_off)
{
	if constexpr (!Q) {
		lo ^= (c << 4) ^ r;
	} else {
		lo = ~lo;
		hi = ~hi ^ (((c << 4) ^ r) << 24);
	}
	const uint2 t0 = lane_lookup<Q, 0>(tab, lo, hi, lane_off), t1 = lane_lookup<Q, 1>(tab, lo, hi, lane_off);
	const uint2 t2 = lane_lookup<Q, 2>(tab, lo, hi, lane_off), t3 = lane_lookup<Q, 3>(tab, lo, hi, lane_off);
	const uint2 t4 = lane_lookup<Q, 4>(tab, lo, hi, lane_off), t5 = lane_lookup<Q, 5>(tab, lo, hi, lane_off);
	const uint2 t6 = lane_lookup<Q, 6>(tab, lo, hi, lane_off), t7 = lane_lookup<Q, 7>(tab, lo, hi, lane_off);
	const uint2 o = mix_rows(t0, t1, t2, t3, t4, t5, t6, t7);
	lo = o.x;
	hi = o.y;
}

__device__ __forceinline__ void perm_p_lanes(uint32_t &lo, uint32_t &hi, uint32_t c, const char *tab, uint32_t lane_off)
{
#pragma unroll 1
	for (uint32_t r = 0; r < 10; r++) round_lanes<false>(lo, hi, c, r, tab, lane_off);
}

// h <- P(h ^ m) ^ Q(m) ^ h on eight lanes
__device__ __forceinline__ void compress_lanes(uint2 &h, uint2 m, uint32_t c, const char *tab, uint32_t lane_off)
{
	uint32_t plo = h.x ^ m.x, phi = h.y ^ m.y, qlo = m.x, qhi = m.y;
#pragma unroll 1
	for (uint32_t r = 0; r < 10; r++) {
		round_lanes<false>(plo, phi, c, r, tab, lane_off);
		round_lanes<true>(qlo, qhi, c, r, tab, lane_off);
	}
	h.x ^= plo ^ qlo;
	h.y ^= phi ^ qhi;
}

constexpr int kLaneThreads = 256; // 32 hashes per workgroup

// Latency form of k_groestl_leaves for small leaf counts (eight lanes per leaf): the same digests.
__global__ __launch_bounds__(kLaneThreads) void k_groestl_leaves_lanes(const uint2 *__restrict__ elems, uint64_t batch, uint64_t n_leaves,
                                                                       uint2 *__restrict__ digests)
{
	extern __shared__ __align__(16) unsigned char smem[];
	require_table_at_lds_zero(smem);
	stage_table(reinterpret_cast<uint2 *>(smem));
	const char *tab = reinterpret_cast<const char *>(smem);
	const uint32_t lane_off = (threadIdx.x & (kCopies - 1)) * 8;
	const uint32_t l16 = threadIdx.x & 15, c = l16 >> 1;
	const uint64_t leaf_raw = (uint64_t)blockIdx.x * (kLaneThreads / 8) + (threadIdx.x >> 4) * 2 + (l16 & 1);
	const bool act = leaf_raw < n_leaves;
	const uint64_t leaf = act ? leaf_raw : 0; // (idle slots recompute leaf 0: every lane of a row stays in step for the DPP)
	const uint64_t n_full = batch >> 2;
	const uint32_t rem = (uint32_t)(batch & 3);
	const uint2 *src = elems + leaf * batch * 2 + c; // column c of block b at src[8 * b]
	uint2 h{0, c == 7 ? 0x00010000u : 0u};
	for (uint64_t b = 0; b < n_full; b++) compress_lanes(h, src[8 * b], c, tab, lane_off);
	{
		uint2 m{0, 0};
		if (c < 2 * rem) m = src[8 * n_full];
		if (c == 2 * rem) m.x = 0x80;
		if (c == 7) m.y = __builtin_bswap32((uint32_t)(n_full + 1));
		compress_lanes(h, m, c, tab, lane_off);
	}
	uint32_t lo = h.x, hi = h.y;
	perm_p_lanes(lo, hi, c, tab, lane_off);
	if (act && c >= 4) digests[4 * leaf + (c - 4)] = uint2{lo ^ h.x, hi ^ h.y};
}

// Latency form of k_groestl_layer.
__global__ __launch_bounds__(kLaneThreads) void k_groestl_layer_lanes(const uint2 *__restrict__ prev, uint64_t n_out, uint2 *__restrict__ next)
{
	extern __shared__ __align__(16) unsigned char smem[];
	require_table_at_lds_zero(smem);
	stage_table(reinterpret_cast<uint2 *>(smem));
	const char *tab = reinterpret_cast<const char *>(smem);
	const uint32_t lane_off = (threadIdx.x & (kCopies - 1)) * 8;
	const uint32_t l16 = threadIdx.x & 15, c = l16 >> 1;
	const uint64_t i_raw = (uint64_t)blockIdx.x * (kLaneThreads / 8) + (threadIdx.x >> 4) * 2 + (l16 & 1);
	const bool act = i_raw < n_out;
	const uint64_t i = act ? i_raw : 0;
	const uint2 x = prev[8 * i + c];
	uint32_t lo = x.x, hi = x.y;
	perm_p_lanes(lo, hi, c, tab, lane_off);
	if (act && c >= 4) next[4 * i + (c - 4)] = uint2{lo ^ x.x, hi ^ x.y};
}

constexpr int kTopThreads = 1024;
constexpr int kTopMaxIn = 1024; // digests of the widest layer it takes

// Workgroup b takes the n_sub (<= kTopMaxIn) consecutive digests b * n_sub .. of a layer that is n_total
// digests wide and walks `levels` (<= log2 n_sub) levels up; level j of the whole tree is
// n_total >> j digests wide and starts right behind level j - 1 (the flattened order), workgroup b owns
// its digests b * (n_sub >> j) ...  One workgroup with n_sub = n_total is the top of the tree.
__global__ __launch_bounds__(kTopThreads) void k_groestl_top(uint4 *__restrict__ layer, uint64_t n_total, uint32_t n_sub, uint32_t levels)
{
	extern __shared__ __align__(16) unsigned char smem[];
	require_table_at_lds_zero(smem);
	stage_table(reinterpret_cast<uint2 *>(smem));
	const char *tab = reinterpret_cast<const char *>(smem);
	uint2 *buf_a = reinterpret_cast<uint2 *>(smem + kTableBytes); // kTopMaxIn digests = 32 KiB
	uint2 *buf_b = buf_a + 4 * kTopMaxIn;                          // kTopMaxIn / 2 digests = 16 KiB
	const uint32_t lane_off = (threadIdx.x & (kCopies - 1)) * 8;
	{
		uint4 *a4 = reinterpret_cast<uint4 *>(buf_a);
		const uint4 *in = layer + 2 * (uint64_t)blockIdx.x * n_sub;
		for (uint32_t i = threadIdx.x; i < 2 * n_sub; i += kTopThreads) a4[i] = in[i];
	}
	__syncthreads();
	const uint32_t l16 = threadIdx.x & 15, c = l16 >> 1;
	const uint32_t slot = (threadIdx.x >> 4) * 2 + (l16 & 1); // hash slot of this lane: kTopThreads / 8 per pass
	uint2 *level = reinterpret_cast<uint2 *>(layer + 2 * n_total); // level 1 of the whole layer
	uint64_t width = n_total >> 1;
	uint2 *src = buf_a, *dst = buf_b;
	uint32_t n = n_sub >> 1;
	for (uint32_t lv = 0; lv < levels; lv++, n >>= 1) {
		uint2 *out = level + 4 * (uint64_t)blockIdx.x * n;
		for (uint32_t base = 0; base < n; base += kTopThreads / 8) {
			if (base + (threadIdx.x >> 4) * 2 >= n) break; // (a 16-lane row leaves or stays as a whole)
			const uint32_t p = base + slot;
			const bool act = p < n;
			const uint32_t pc = act ? p : 0;
			const uint2 x = src[8 * pc + c]; // column c of child(2p) || child(2p + 1)
			uint32_t lo = x.x, hi = x.y;
			perm_p_lanes(lo, hi, c, tab, lane_off);
			if (act && c >= 4) { // Omega: columns 4..7 of P(x) ^ x
				const uint2 d{lo ^ x.x, hi ^ x.y};
				dst[4 * p + (c - 4)] = d;
				out[4 * p + (c - 4)] = d;
			}
		}
		__syncthreads();
		level += 4 * width;
		width >>= 1;
		uint2 *t = src;
		src = dst;
		dst = t;
	}
}

} // namespace

static hipError_t set_lds_limits()
{
	hipError_t e = func_lds_limit(reinterpret_cast<const void *>(k_groestl_leaves), (int)kTableBytes);
	if (e != hipSuccess) return e;
	e = func_lds_limit(reinterpret_cast<const void *>(k_groestl_layer), (int)kTableBytes);
	if (e != hipSuccess) return e;
	e = func_lds_limit(reinterpret_cast<const void *>(k_groestl_leaves_lanes), (int)kTableBytes);
	if (e != hipSuccess) return e;
	e = func_lds_limit(reinterpret_cast<const void *>(k_groestl_layer_lanes), (int)kTableBytes);
	if (e != hipSuccess) return e;
	return func_lds_limit(reinterpret_cast<const void *>(k_groestl_top), (int)(kTableBytes + 48 * kTopMaxIn));
}

// up to this many hashes a launch uses the eight-lanes-per-hash kernels (BN_GROESTL_LANES_MAX overrides)
static uint64_t lanes_max()
{
	static const uint64_t v = [] {
		const char *e = getenv("BN_GROESTL_LANES_MAX");
		return e ? (uint64_t)strtoull(e, nullptr, 10) : (uint64_t)1 << 16;
	}();
	return v;
}

static unsigned grid_for(uint64_t n, int n_cu)
{
	uint64_t blocks = (n + kThreads - 1) / kThreads;
	const uint64_t cap = (uint64_t)n_cu * 2; // two 64 KiB tables per CU
	if (blocks > cap) blocks = cap;
	return (unsigned)(blocks ? blocks : 1);
}

hipError_t launch_groestl_leaves(hipStream_t s, int n_cu, const void *elems, uint64_t batch, uint64_t n_leaves, void *digests)
{
	if (n_leaves == 0) return hipSuccess;
	hipError_t e = set_lds_limits();
	if (e != hipSuccess) return e;
	if (n_leaves <= lanes_max()) { // latency form: eight lanes per hash, ~7x shorter dependency chain
		const unsigned blocks = (unsigned)((n_leaves + kLaneThreads / 8 - 1) / (kLaneThreads / 8));
		hipLaunchKernelGGL(k_groestl_leaves_lanes, dim3(blocks), dim3(kLaneThreads), kTableBytes, s, (const uint2 *)elems, batch, n_leaves, (uint2 *)digests);
		return hipGetLastError();
	}
	// (two whole-block leaves per lane at a time, so that the lone P permutations at the end of a hash advance in pairs, was measured
	// and is not kept: 0.673 against 0.681 ms for 2^20 leaves of 256 bytes -- the kernel is bound by throughput, not by the chain)
	hipLaunchKernelGGL(k_groestl_leaves, dim3(grid_for(n_leaves, n_cu)), dim3(kThreads), kTableBytes, s, (const uint4 *)elems, batch, n_leaves,
	                   (uint4 *)digests);
	return hipGetLastError();
}

hipError_t launch_groestl_layer(hipStream_t s, int n_cu, const void *prev, uint64_t n_out, void *next)
{
	if (n_out == 0) return hipSuccess;
	hipError_t e = set_lds_limits();
	if (e != hipSuccess) return e;
	if (n_out <= lanes_max()) {
		const unsigned blocks = (unsigned)((n_out + kLaneThreads / 8 - 1) / (kLaneThreads / 8));
		hipLaunchKernelGGL(k_groestl_layer_lanes, dim3(blocks), dim3(kLaneThreads), kTableBytes, s, (const uint2 *)prev, n_out, (uint2 *)next);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_groestl_layer, dim3(grid_for(n_out, n_cu)), dim3(kThreads), kTableBytes, s, (const uint4 *)prev, n_out, (uint4 *)next);
	return hipGetLastError();
}

// out[i * item_elems + e] = src[offsets[i] + e]: the openings of a committed vector (branch digests,
// cosets).  offsets and out live in pinned host memory mapped into the device (zero-copy both ways).
__global__ void k_gather(const uint4 *__restrict__ src, const uint64_t *__restrict__ offsets, uint64_t n_items, uint64_t item_elems,
                         uint4 *__restrict__ out)
{
	const uint64_t total = n_items * item_elems;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t item = t / item_elems, e = t - item * item_elems;
		out[t] = src[offsets[item] + e];
	}
}

hipError_t launch_gather(hipStream_t s, const void *src, const uint64_t *offsets, uint64_t n_items, uint64_t item_elems, void *out)
{
	const uint64_t total = n_items * item_elems;
	if (total == 0) return hipSuccess;
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 1024) blocks = 1024;
	hipLaunchKernelGGL(k_gather, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4 *)src, offsets, n_items, item_elems, (uint4 *)out);
	return hipGetLastError();
}

// nodes: the flattened tree (leaf digests already at the front).  Layers wider than 2^20 digests one
// throughput launch each; below that, sub-trees of 1024 digests per workgroup (ten levels per launch:
// 7 G compressions/s with one launch against ~5 G/s and ten launches of the per-layer kernels at
// these widths), the last one being the top of the tree.
hipError_t launch_merkle_layers(hipStream_t s, int n_cu, void *nodes, uint64_t n_leaves)
{
	hipError_t e = set_lds_limits();
	if (e != hipSuccess) return e;
	static const uint64_t subtree_max = [] {
		const char *v = getenv("BN_GROESTL_SUBTREE_MAX_LOG2");
		return (uint64_t)1 << (v ? atoi(v) : 20);
	}();
	char *layer = (char *)nodes;
	uint64_t n = n_leaves;
	while (n > subtree_max && n > (uint64_t)kTopMaxIn) {
		char *next = layer + 32 * n;
		e = launch_groestl_layer(s, n_cu, layer, n >> 1, next);
		if (e != hipSuccess) return e;
		layer = next;
		n >>= 1;
	}
	while (n >= 2) {
		const uint32_t n_sub = n < (uint64_t)kTopMaxIn ? (uint32_t)n : (uint32_t)kTopMaxIn;
		uint32_t levels = 0;
		while ((1u << levels) < n_sub) levels++;
		hipLaunchKernelGGL(k_groestl_top, dim3((unsigned)(n / n_sub)), dim3(kTopThreads), kTableBytes + 48 * kTopMaxIn, s, (uint4 *)layer, n, n_sub,
		                   levels);
		e = hipGetLastError();
		if (e != hipSuccess) return e;
		for (uint32_t j = 0; j < levels; j++) {
			layer += 32 * n;
			n >>= 1;
		}
	}
	return hipSuccess;
}

} // namespace bn
