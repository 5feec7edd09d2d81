"""Pins oracle/merkle_ref.c (Groestl-256, the 2-to-1 Groestl compression, the binary Merkle tree):
the published Groestl-256 known answers, an independent table-driven Python Groestl at every padding
boundary, and the Merkle tree's structural identities (binary_merkle_tree.rs:22-25, :118-141)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def oracle():
    import oracle as o

    o.build()
    return o


def test_published_known_answers(oracle):
    kat = json.load(open(os.path.join(HERE, "golden", "groestl256_kat.json")))
    assert len(kat["vectors"]) >= 4
    for v in kat["vectors"]:
        assert oracle.groestl256(v["msg_utf8"].encode()).hex() == v["digest"]


# ---- an independent formulation: 64-bit column tables (SubBytes + MixBytes fused), as software
# implementations of the specification's section 6 do it
def _sbox():
    s, p, q = [0] * 256, 1, 1
    while True:
        p = (p ^ (p << 1) ^ (0x1B if p & 0x80 else 0)) & 0xFF
        q ^= (q << 1) & 0xFF
        q ^= (q << 2) & 0xFF
        q ^= (q << 4) & 0xFF
        if q & 0x80:
            q ^= 0x09
        r = lambda x, k: ((x << k) | (x >> (8 - k))) & 0xFF
        s[p] = q ^ r(q, 1) ^ r(q, 2) ^ r(q, 3) ^ r(q, 4) ^ 0x63
        if p == 1:
            break
    s[0] = 0x63
    return s


def _gmul(x, m):
    acc = 0
    for _ in range(3):
        if m & 1:
            acc ^= x
        m >>= 1
        x = ((x << 1) ^ (0x1B if x & 0x80 else 0)) & 0xFF
    return acc


_S = _sbox()
_CIRC = [2, 2, 3, 4, 5, 3, 5, 7]
# T[k][b]: the contribution of a byte b sitting in row k of a column to that column's 8 output rows
_T = [[[_gmul(_S[b], _CIRC[(k - r) % 8]) for r in range(8)] for b in range(256)] for k in range(8)]


def _perm(cols, is_q):
    sh = [1, 3, 5, 7, 0, 2, 4, 6] if is_q else list(range(8))
    for rnd in range(10):
        for c in range(8):
            if is_q:
                cols[c] = [x ^ 0xFF for x in cols[c]]
                cols[c][7] ^= (c << 4) ^ rnd
            else:
                cols[c][0] ^= (c << 4) ^ rnd
        new = []
        for c in range(8):
            out = [0] * 8
            for k in range(8):
                t = _T[k][cols[(c + sh[k]) % 8][k]]
                out = [a ^ b for a, b in zip(out, t)]
            new.append(out)
        cols = new
    return cols


def _to_cols(b):
    return [list(b[8 * c : 8 * c + 8]) for c in range(8)]


def _py_groestl256(msg):
    h = bytearray(64)
    h[62] = 1
    n_blocks = (len(msg) + 9 + 63) // 64
    padded = bytes(msg) + b"\x80" + bytes(64 * n_blocks - len(msg) - 9) + n_blocks.to_bytes(8, "big")
    for i in range(n_blocks):
        m = padded[64 * i : 64 * i + 64]
        p = _perm(_to_cols(bytes(a ^ b for a, b in zip(h, m))), False)
        q = _perm(_to_cols(m), True)
        for c in range(8):
            for r in range(8):
                h[8 * c + r] ^= p[c][r] ^ q[c][r]
    p = _perm(_to_cols(bytes(h)), False)
    out = bytes(h[8 * c + r] ^ p[c][r] for c in range(8) for r in range(8))
    return out[32:]


@pytest.mark.parametrize("length", [0, 1, 16, 32, 55, 56, 57, 63, 64, 65, 119, 120, 128, 256, 300])
def test_matches_independent_table_formulation(oracle, length):
    msg = bytes((7 * i + 3) & 0xFF for i in range(length))
    assert oracle.groestl256(msg) == _py_groestl256(msg)


def test_two_to_one_compression_is_the_output_transformation(oracle):
    """compression.rs:21-36: last 32 bytes of P(x) ^ x, x = in0 || in1."""
    rng = np.random.default_rng(3)
    for _ in range(4):
        a, b = rng.bytes(32), rng.bytes(32)
        x = a + b
        p = _perm(_to_cols(x), False)
        want = bytes(x[8 * c + r] ^ p[c][r] for c in range(8) for r in range(8))[32:]
        assert oracle.groestl256_compress2(a, b) == want


@pytest.mark.parametrize("n_elems,batch", [(1, 1), (16, 1), (64, 4), (64, 64), (256, 16), (32, 2)])
def test_merkle_tree_structure(oracle, n_elems, batch):
    elems = oracle.random_b128(0x3E51 + n_elems, n_elems)
    rc, nodes = oracle.merkle_build(elems, batch)
    assert rc == 0
    n_leaves = n_elems // batch
    assert nodes.shape == (2 * n_leaves - 1, 32)
    raw = elems.tobytes()  # 16 little-endian bytes per element = the canonical serialization
    for i in range(n_leaves):
        assert bytes(nodes[i]) == oracle.groestl256(raw[16 * batch * i : 16 * batch * (i + 1)])
    # layer(depth) (binary_merkle_tree.rs:109-116) and the parent relation
    log_len = n_leaves.bit_length() - 1
    total = 2 * n_leaves - 1
    for depth in range(log_len):
        start = total + 1 - (1 << (depth + 1))
        below = total + 1 - (1 << (depth + 2))
        for i in range(1 << depth):
            assert bytes(nodes[start + i]) == oracle.groestl256_compress2(bytes(nodes[below + 2 * i]), bytes(nodes[below + 2 * i + 1]))
    # branch(index, 0) (:121-141) verifies against the root the way verify_opening does (scheme.rs:118-146)
    for index in {0, n_leaves - 1, n_leaves // 3}:
        branch = [bytes(nodes[(((1 << j) - 1) << (log_len + 1 - j)) | ((index >> j) ^ 1)]) for j in range(log_len)]
        assert oracle.merkle_root_from_branch(bytes(nodes[index]), index, branch) == bytes(nodes[-1])


def test_merkle_build_rejects_bad_shapes(oracle):
    elems = oracle.random_b128(1, 24)
    assert oracle.merkle_build(elems, 5)[0] == -1  # IncorrectBatchSize (binary_merkle_tree.rs:39-41)
    assert oracle.merkle_build(elems, 8)[0] == -2  # PowerOfTwoLengthRequired (:45-47)
