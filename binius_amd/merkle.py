"""Host-side mirror of the reference's Merkle-tree vector commitment prover, with the tree built on
the device (bn_merkle_build).

Mirrors crates/core/src/merkle_tree/prover.rs:19-106 (BinaryMerkleTreeProver: commit / layer /
prove_opening) and binary_merkle_tree.rs:103-141 (BinaryMerkleTree: root / layer / branch) with
H = Groestl256 and C = Groestl256ByteCompression -- the instantiation every prover entry point of the
reference uses (e.g. crates/core/src/constraint_system/prove.rs, examples).  The committed data and
the node array stay on the device; the root, a layer or a branch is read back when asked for
(bn_gather_d2h).
"""
import numpy as np

from ._ffi import BN_ERR_INPUT_VALIDATION, BnError


class MerkleError(BnError):
    pass


class BinaryMerkleTree:
    """binary_merkle_tree.rs:20-25: log_len + inner_nodes (flattened layers, root last) -- with the node
    array resident on the device: root / layer / branch read back only what they return."""

    def __init__(self, hal, log_len, nodes):
        self.hal = hal
        self.log_len = log_len
        self.nodes = nodes  # device slice of 2 * (2^(log_len+1) - 1) elements
        self._n_nodes = (2 << log_len) - 1

    @staticmethod
    def _digests(a):
        return np.ascontiguousarray(a).view(np.uint8).reshape(-1, 32)

    @property
    def inner_nodes(self):
        """The whole flattened tree on the host ((2^(log_len+1) - 1, 32) uint8): tests and small trees."""
        return self._digests(self.hal.copy_d2h(self.nodes))

    def root(self):
        return bytes(self._digests(self.hal.gather_d2h(self.nodes, [2 * (self._n_nodes - 1)], 2))[0])

    def layer(self, layer_depth):
        if layer_depth > self.log_len:
            raise MerkleError(BN_ERR_INPUT_VALIDATION, "IncorrectLayerDepth")
        start = self._n_nodes + 1 - (1 << (layer_depth + 1))
        return self._digests(self.hal.copy_d2h(self.nodes.slice(2 * start, 2 * (start + (1 << layer_depth)))))

    def _branch_nodes(self, index, layer_depth):
        if index >= (1 << self.log_len) or layer_depth > self.log_len:
            raise MerkleError(BN_ERR_INPUT_VALIDATION, "IndexOutOfRange { max: %d }" % ((1 << self.log_len) - 1))
        return [(((1 << j) - 1) << (self.log_len + 1 - j)) | ((index >> j) ^ 1) for j in range(self.log_len - layer_depth)]

    def branch(self, index, layer_depth):
        return self.branches([index], layer_depth)[0]

    def branches(self, indices, layer_depth):
        """Merkle branches of several leaves with one device gather."""
        node_ids = [self._branch_nodes(i, layer_depth) for i in indices]
        flat = [2 * n for ids in node_ids for n in ids]
        got = self._digests(self.hal.gather_d2h(self.nodes, flat, 2)) if flat else np.zeros((0, 32), np.uint8)
        out, k = [], 0
        for ids in node_ids:
            out.append([bytes(got[k + j]) for j in range(len(ids))])
            k += len(ids)
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
        tree = BinaryMerkleTree(self.hal, log_len, nodes)
        return (tree.root(), tree.log_len), tree

    def layer(self, committed, depth):
        return committed.layer(depth)

    def prove_opening(self, committed, layer_depth, index):
        """The branch the reference writes to the transcript (prover.rs:73-83)."""
        return committed.branch(index, layer_depth)
