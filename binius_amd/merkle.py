"""Host-side mirror of the reference's Merkle-tree vector commitment prover, with the tree built on
the device (bn_merkle_build).

Mirrors crates/core/src/merkle_tree/prover.rs:19-106 (BinaryMerkleTreeProver: commit / layer /
prove_opening) and binary_merkle_tree.rs:103-141 (BinaryMerkleTree: root / layer / branch) with
H = Groestl256 and C = Groestl256ByteCompression -- the instantiation every prover entry point of the
reference uses (e.g. crates/core/src/constraint_system/prove.rs, examples).  The committed data stays
on the device; only the 32-byte nodes come back.
"""
import numpy as np

from ._ffi import BN_ERR_INPUT_VALIDATION, BnError


class MerkleError(BnError):
    pass


class BinaryMerkleTree:
    """binary_merkle_tree.rs:20-25: log_len + inner_nodes (flattened layers, root last)."""

    def __init__(self, log_len, inner_nodes):
        self.log_len = log_len
        self.inner_nodes = inner_nodes  # (2^(log_len+1) - 1, 32) uint8

    def root(self):
        return bytes(self.inner_nodes[-1])

    def layer(self, layer_depth):
        if layer_depth > self.log_len:
            raise MerkleError(BN_ERR_INPUT_VALIDATION, "IncorrectLayerDepth")
        start = len(self.inner_nodes) + 1 - (1 << (layer_depth + 1))
        return self.inner_nodes[start : start + (1 << layer_depth)]

    def branch(self, index, layer_depth):
        if index >= (1 << self.log_len) or layer_depth > self.log_len:
            raise MerkleError(BN_ERR_INPUT_VALIDATION, "IndexOutOfRange { max: %d }" % ((1 << self.log_len) - 1))
        out = []
        for j in range(self.log_len - layer_depth):
            node_index = (((1 << j) - 1) << (self.log_len + 1 - j)) | ((index >> j) ^ 1)
            out.append(bytes(self.inner_nodes[node_index]))
        return out


class BinaryMerkleTreeProver:
    """prover.rs:19-106.  `hal` is a binius_amd ComputeLayer, `dev_alloc` a device bump allocator that
    the flattened tree (2 * (2 * n_leaves - 1) elements) is taken from."""

    def __init__(self, hal, dev_alloc):
        self.hal = hal
        self.dev_alloc = dev_alloc

    def commit(self, data, batch_size):
        """data: device slice of BinaryField128b.  Returns ((root, depth), BinaryMerkleTree)."""
        if batch_size == 0 or data.len % batch_size != 0:
            raise MerkleError(BN_ERR_INPUT_VALIDATION, "IncorrectBatchSize")
        n_leaves = data.len // batch_size
        if n_leaves & (n_leaves - 1):
            raise MerkleError(BN_ERR_INPUT_VALIDATION, "PowerOfTwoLengthRequired")
        log_len = n_leaves.bit_length() - 1
        nodes = self.dev_alloc.alloc(2 * (2 * n_leaves - 1))
        self.hal.merkle_build(data, batch_size, nodes)
        host = self.hal.copy_d2h(nodes)
        tree = BinaryMerkleTree(log_len, np.ascontiguousarray(host).view(np.uint8).reshape(-1, 32))
        return (tree.root(), tree.log_len), tree

    def layer(self, committed, depth):
        return committed.layer(depth)

    def prove_opening(self, committed, layer_depth, index):
        """The branch the reference writes to the transcript (prover.rs:73-83)."""
        return committed.branch(index, layer_depth)
