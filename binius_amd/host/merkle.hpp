// binius_amd/host/merkle.hpp -- C++ mirror of the reference's Merkle-tree vector commitment prover
// with the tree built on the device (bn_merkle_build, include/binius_amd.h).
//
// Mirrors, with H = Groestl256 and C = Groestl256ByteCompression:
//   BinaryMerkleTree{log_len, inner_nodes}, root / layer / branch   binary_merkle_tree.rs:20-25, 103-141
//   BinaryMerkleTreeProver::commit / layer / prove_opening           prover.rs:19-106
// The caller of the reference (FRIFolder::execute_fold_round, fri/prove.rs:395-420) copies the folded
// codeword to the host and hashes it there; with this class the codeword stays on the device and only
// the 32-byte nodes come back.
#pragma once
#include <array>
#include <cstring>
#include <vector>

#include "compute_layer.hpp"

namespace binius_amd {

using Digest = std::array<uint8_t, 32>; // digest::Output<Groestl256>

class MerkleError : public Error {
public:
	explicit MerkleError(const std::string &what) : Error(InputValidation, what) {}
};

struct Commitment { // merkle_tree_vcs.rs: Commitment{root, depth}
	Digest root;
	size_t depth;
};

class BinaryMerkleTree {
public:
	size_t log_len = 0;
	std::vector<Digest> inner_nodes; // flattened layers, leaves first, root last

	Digest root() const { return inner_nodes.back(); }
	// binary_merkle_tree.rs:109-116
	std::pair<const Digest *, size_t> layer(size_t layer_depth) const
	{
		if (layer_depth > log_len) throw MerkleError("IncorrectLayerDepth");
		const size_t start = inner_nodes.size() + 1 - ((size_t)1 << (layer_depth + 1));
		return {inner_nodes.data() + start, (size_t)1 << layer_depth};
	}
	// binary_merkle_tree.rs:121-141
	std::vector<Digest> branch(size_t index, size_t layer_depth) const
	{
		if (index >= ((size_t)1 << log_len) || layer_depth > log_len)
			throw MerkleError("IndexOutOfRange { max: " + std::to_string(((size_t)1 << log_len) - 1) + " }");
		std::vector<Digest> out;
		for (size_t j = 0; j < log_len - layer_depth; j++) {
			const size_t node_index = ((((size_t)1 << j) - 1) << (log_len + 1 - j)) | ((index >> j) ^ 1);
			out.push_back(inner_nodes[node_index]);
		}
		return out;
	}
};

class BinaryMerkleTreeProver {
public:
	explicit BinaryMerkleTreeProver(ComputeLayer &hal) : hal_(hal) {}

	// device memory the flattened tree of `n_leaves` leaves needs, in field elements
	static size_t required_device_memory(size_t n_leaves) { return 2 * (2 * n_leaves - 1); }

	// prover.rs:47-62.  `data` is a device slice; `dev_alloc` provides the node array.
	std::pair<Commitment, BinaryMerkleTree> commit(FSlice data, size_t batch_size, DeviceBumpAllocator &dev_alloc)
	{
		if (batch_size == 0 || data.len() % batch_size != 0) throw MerkleError("IncorrectBatchSize");
		const size_t n_leaves = data.len() / batch_size;
		if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) != 0) throw MerkleError("PowerOfTwoLengthRequired");
		FSliceMut nodes = dev_alloc.alloc(required_device_memory(n_leaves));
		check(bn_merkle_build(hal_.raw_ctx(), data.ptr, data.len(), batch_size, nodes.ptr));
		BinaryMerkleTree tree;
		while (((size_t)1 << tree.log_len) < n_leaves) tree.log_len++;
		tree.inner_nodes.resize(2 * n_leaves - 1);
		static_assert(sizeof(Digest) == 2 * sizeof(B128), "a digest is two field elements wide");
		hal_.copy_d2h(ComputeMemory::as_const(nodes), reinterpret_cast<B128 *>(tree.inner_nodes.data()), nodes.len());
		return {Commitment{tree.root(), tree.log_len}, std::move(tree)};
	}
	std::pair<const Digest *, size_t> layer(const BinaryMerkleTree &committed, size_t depth) const { return committed.layer(depth); }
	// prover.rs:73-83: the branch that is written to the transcript
	std::vector<Digest> prove_opening(const BinaryMerkleTree &committed, size_t layer_depth, size_t index) const
	{
		return committed.branch(index, layer_depth);
	}

private:
	ComputeLayer &hal_;
};

} // namespace binius_amd
