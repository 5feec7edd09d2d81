#!/usr/bin/env python3
"""Timing of the device Merkle commitment (bn_merkle_build) on a 2^log_n-element BinaryField128b
codeword, per batch size (= FRI coset size, crates/core/src/protocols/fri/prove.rs:400-407), with the
leaf-hash and layer kernels also timed alone.  One JSON line per case; inputs from binius_amd.synthetic.
Groestl is compute-bound: the rate to look at is message bytes hashed per second, not the HBM roofline
(DESIGN.md section 4.8)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=24)
ap.add_argument("--batches", type=int, nargs="*", default=[1, 4, 16, 64])
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n = 1 << a.log_n
hal = binius_amd.Context(0, n + 4 * n + (1 << 12))
alloc = hal.dev_alloc()
data = alloc.alloc(n)
hal.copy_h2d(synthetic.random_b128(0xB1A5, n), data)
nodes_all = alloc.alloc(4 * n)


def timed(fn):
    ts = []
    for _ in range(a.reps + 1):
        hal.sync(); hal.timer_begin(); fn(); ts.append(hal.timer_end_ms())
    return min(ts[1:])


for batch in a.batches:
    n_leaves = n // batch
    nodes = nodes_all.slice(0, 2 * (2 * n_leaves - 1))
    leaves = nodes.slice(0, 2 * n_leaves)
    t_all = timed(lambda: hal.merkle_build(data, batch, nodes))
    t_leaf = timed(lambda: hal.groestl256_leaves(data, batch, leaves))
    nxt = nodes.slice(2 * n_leaves, 3 * n_leaves)
    t_layer = timed(lambda: hal.groestl256_compress_layer(leaves, nxt)) if n_leaves >= 2 else 0.0
    # permutations: a leaf = 2 * (full blocks + 1 padding block) + 1 (output transformation) -- one fewer where the leaf is whole blocks
    # (batch % 4 == 0: Q of the constant padding block is computed once per lane, not per leaf); a tree node = 1
    perms = n_leaves * (2 * (batch // 4 + 1) + 1 - (1 if batch % 4 == 0 else 0)) + (n_leaves - 1)
    print(json.dumps({
        "op": "merkle_build 2^%d elems, batch %d (2^%d leaves)" % (a.log_n, batch, n_leaves.bit_length() - 1),
        "ms": round(t_all, 4), "leaves_ms": round(t_leaf, 4), "first_layer_ms": round(t_layer, 4),
        "codeword_GBps": round(16 * n / t_all / 1e6, 1),
        "G_permutations_per_s": round(perms / t_all / 1e6, 2),
    }), flush=True)
