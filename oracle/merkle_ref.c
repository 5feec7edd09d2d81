/* oracle/merkle_ref.c -- see merkle_ref.h.  TEST INFRASTRUCTURE ONLY.
 * Byte-level Groestl-256 written from the specification: no tables beyond the AES S-box, which is
 * itself generated from its definition (inverse in GF(2^8) mod x^8+x^4+x^3+x+1, then the affine map). */
#include "merkle_ref.h"

#include <string.h>

static uint8_t SBOX[256];
static int sbox_ready;

static uint8_t rotl8(uint8_t x, int s) { return (uint8_t)((x << s) | (x >> (8 - s))); }

static void sbox_init(void)
{
	/* p walks the multiplicative group by *3, q by /3, so q = p^-1 throughout */
	uint8_t p = 1, q = 1;
	do {
		p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1B : 0));
		q ^= (uint8_t)(q << 1);
		q ^= (uint8_t)(q << 2);
		q ^= (uint8_t)(q << 4);
		if (q & 0x80) q ^= 0x09;
		SBOX[p] = (uint8_t)(q ^ rotl8(q, 1) ^ rotl8(q, 2) ^ rotl8(q, 3) ^ rotl8(q, 4) ^ 0x63);
	} while (p != 1);
	SBOX[0] = 0x63;
	sbox_ready = 1;
}

static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x & 0x80) ? 0x1B : 0)); }

/* The state is an 8 x 8 byte matrix; byte i of a 64-byte block is row i % 8 of column i / 8
 * (specification section 3.4.1). s[col][row]. */
typedef uint8_t state_t[8][8];

static const int SHIFT_P[8] = {0, 1, 2, 3, 4, 5, 6, 7};
static const int SHIFT_Q[8] = {1, 3, 5, 7, 0, 2, 4, 6};

static void permutation(state_t s, int is_q)
{
	for (int round = 0; round < 10; round++) {
		/* AddRoundConstant (3.4.2) */
		if (!is_q) {
			for (int c = 0; c < 8; c++) s[c][0] ^= (uint8_t)((c << 4) ^ round);
		} else {
			for (int c = 0; c < 8; c++) {
				for (int r = 0; r < 7; r++) s[c][r] ^= 0xFF;
				s[c][7] ^= (uint8_t)(0xFF ^ (c << 4) ^ round);
			}
		}
		/* SubBytes (3.4.3) */
		for (int c = 0; c < 8; c++)
			for (int r = 0; r < 8; r++) s[c][r] = SBOX[s[c][r]];
		/* ShiftBytes (3.4.4): row r moves left by sigma_r columns */
		state_t t;
		const int *sh = is_q ? SHIFT_Q : SHIFT_P;
		for (int c = 0; c < 8; c++)
			for (int r = 0; r < 8; r++) t[c][r] = s[(c + sh[r]) & 7][r];
		/* MixBytes (3.4.5): every column times circ(02, 02, 03, 04, 05, 03, 05, 07) */
		static const uint8_t B[8] = {2, 2, 3, 4, 5, 3, 5, 7};
		for (int c = 0; c < 8; c++)
			for (int r = 0; r < 8; r++) {
				uint8_t acc = 0;
				for (int k = 0; k < 8; k++) {
					const uint8_t x = t[c][(r + k) & 7], x2 = xtime(x), x4 = xtime(x2);
					const uint8_t m = B[k];
					acc ^= (uint8_t)(((m & 1) ? x : 0) ^ ((m & 2) ? x2 : 0) ^ ((m & 4) ? x4 : 0));
				}
				s[c][r] = acc;
			}
	}
}

/* f(h, m) = P(h ^ m) ^ Q(m) ^ h  (3.2; crates/hash/src/groestl/mod.rs:26-34) */
static void compress(uint8_t h[64], const uint8_t m[64])
{
	state_t p, q;
	for (int i = 0; i < 64; i++) {
		p[i / 8][i % 8] = (uint8_t)(h[i] ^ m[i]);
		q[i / 8][i % 8] = m[i];
	}
	permutation(p, 0);
	permutation(q, 1);
	for (int i = 0; i < 64; i++) h[i] ^= (uint8_t)(p[i / 8][i % 8] ^ q[i / 8][i % 8]);
}

/* Omega(h) = trunc(P(h) ^ h)  (3.3) */
static void output_transform(const uint8_t h[64], uint8_t out[32])
{
	state_t p;
	for (int i = 0; i < 64; i++) p[i / 8][i % 8] = h[i];
	permutation(p, 0);
	for (int i = 32; i < 64; i++) out[i - 32] = (uint8_t)(p[i / 8][i % 8] ^ h[i]);
}

void ref_groestl256(const uint8_t *msg, size_t len, uint8_t out[32])
{
	if (!sbox_ready) sbox_init();
	uint8_t h[64];
	memset(h, 0, 64);
	h[62] = 0x01; /* iv = the output size in bits (256), big-endian, in the last 8 bytes (3.1; digest.rs:67-69) */
	size_t n_blocks = 0;
	while (len >= 64) {
		compress(h, msg);
		msg += 64;
		len -= 64;
		n_blocks++;
	}
	/* padding (3.1): 0x80, zeros, then the total number of blocks as a 64-bit big-endian integer */
	uint8_t last[128];
	memset(last, 0, sizeof last);
	memcpy(last, msg, len);
	last[len] = 0x80;
	const size_t tail_blocks = (len + 1 + 8 <= 64) ? 1 : 2;
	const uint64_t total = (uint64_t)n_blocks + tail_blocks;
	for (int i = 0; i < 8; i++) last[64 * tail_blocks - 1 - i] = (uint8_t)(total >> (8 * i));
	for (size_t b = 0; b < tail_blocks; b++) compress(h, last + 64 * b);
	output_transform(h, out);
}

void ref_groestl256_compress2(const uint8_t in0[32], const uint8_t in1[32], uint8_t out[32])
{
	if (!sbox_ready) sbox_init();
	uint8_t x[64];
	memcpy(x, in0, 32);
	memcpy(x + 32, in1, 32);
	output_transform(x, out);
}

int ref_merkle_build(const uint8_t *elems, uint64_t n_elems, uint64_t batch_size, uint8_t *nodes)
{
	if (batch_size == 0 || n_elems % batch_size != 0) return -1;
	const uint64_t n_leaves = n_elems / batch_size;
	if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) != 0) return -2;
	for (uint64_t i = 0; i < n_leaves; i++)
		ref_groestl256(elems + i * batch_size * 16, (size_t)(batch_size * 16), nodes + 32 * i);
	uint8_t *prev = nodes;
	for (uint64_t w = n_leaves / 2; w >= 1; w /= 2) {
		uint8_t *next = prev + 32 * (2 * w);
		for (uint64_t i = 0; i < w; i++) ref_groestl256_compress2(prev + 64 * i, prev + 64 * i + 32, next + 32 * i);
		prev = next;
	}
	return 0;
}

void ref_merkle_root_from_branch(const uint8_t leaf[32], uint64_t index, const uint8_t *branch, uint32_t depth, uint8_t out[32])
{
	uint8_t cur[32];
	memcpy(cur, leaf, 32);
	for (uint32_t d = 0; d < depth; d++) {
		const uint8_t *sib = branch + 32 * d;
		uint8_t nxt[32];
		if ((index & 1) == 0)
			ref_groestl256_compress2(cur, sib, nxt);
		else
			ref_groestl256_compress2(sib, cur, nxt);
		memcpy(cur, nxt, 32);
		index >>= 1;
	}
	memcpy(out, cur, 32);
}
