/* oracle/merkle_ref.h -- CPU restatement of the Groestl-256 hash, the Groestl output-transformation
 * 2-to-1 compression and the binary Merkle tree of the reference's vector commitment.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Where the algorithm lives.  The reference builds its own Groestl from the published specification
 * (crates/hash/src/groestl/digest.rs:30-87 the hash construction; mod.rs:26-34 the compression
 * function f(h, m) = P(h ^ m) ^ Q(m) ^ h; compression.rs:21-36 the 2-to-1 compression) and tests it
 * only against the third-party crate `groestl` 0.10.1 (Cargo.toml:102, crates/hash/src/groestl/
 * tests.rs:9-18), which is not vendored under /root/reference and cannot be built here (no Rust).
 * The restatement therefore follows the Groestl specification (Gauravaram et al., "Groestl -- a SHA-3
 * candidate", 2011, sections 3.1-3.4) byte by byte, and is pinned by the specification's published
 * known answers (tests/test_oracle_merkle.py, tests/golden/groestl256_kat.json).
 *
 * Merkle tree: crates/core/src/merkle_tree/binary_merkle_tree.rs:27-101 (build / internal_build),
 * :158-168 (compress_layer), :175-211 (hash_interleaved: leaf i = hash of the serialization of
 * elements [i*batch, (i+1)*batch)); serialization of a BinaryField128b = 16 little-endian bytes
 * (crates/utils/src/serialization.rs:94-104). */
#ifndef BN_ORACLE_MERKLE_REF_H
#define BN_ORACLE_MERKLE_REF_H
#include <stddef.h>
#include <stdint.h>

/* Groestl-256 of msg[0 .. len) (digest.rs:62-87: padding with the BLOCK count, truncation to the
 * last 32 bytes of P(h) ^ h). */
void ref_groestl256(const uint8_t *msg, size_t len, uint8_t out[32]);

/* Groestl256ByteCompression (compression.rs:21-36): last 32 bytes of P(x) ^ x, x = in0 || in1. */
void ref_groestl256_compress2(const uint8_t in0[32], const uint8_t in1[32], uint8_t out[32]);

/* binary_merkle_tree::build over `n_elems` 16-byte elements hashed `batch_size` at a time.
 * nodes: (2 * n_leaves - 1) * 32 bytes, layers flattened leaves first, root last (:22-25).
 * returns 0, or -1 (n_elems % batch_size != 0) / -2 (leaf count not a power of two)  (:39-48). */
int ref_merkle_build(const uint8_t *elems, uint64_t n_elems, uint64_t batch_size, uint8_t *nodes);

/* MerkleTreeScheme::verify_opening shape (merkle_tree/scheme.rs): recompute the root of leaf `index`
 * from its digest and the branch (sibling digests, leaf level first).  out = the recomputed root. */
void ref_merkle_root_from_branch(const uint8_t leaf[32], uint64_t index, const uint8_t *branch, uint32_t depth, uint8_t out[32]);

#endif
