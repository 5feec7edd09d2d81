// binius_amd/host/merkle.hpp -- C++ mirror of the reference's Merkle-tree vector commitment prover
// with the tree built on the device (bn_merkle_build, include/binius_amd.h).
//
// Mirrors, with H = Groestl256 and C = Groestl256ByteCompression:
//   BinaryMerkleTree{log_len, inner_nodes}, root / layer / branch   binary_merkle_tree.rs:20-25, 103-141
//   BinaryMerkleTreeProver::commit / layer / prove_opening           prover.rs:19-106
// The caller of the reference (FRIFolder::execute_fold_round, fri/prove.rs:395-420) copies the folded
// codeword to the host and hashes it there; with this class the codeword and the tree stay on the
// device and only roots, layers and branches come back.
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

// The node array lives on the device; root / layer / branch read back what they return
// (bn_gather_d2h), inner_nodes() the whole array.
class BinaryMerkleTree {
public:
	size_t log_len = 0;
	FSlice nodes{};          // 2 * (2^(log_len+1) - 1) elements: flattened layers, leaves first, root last
	ComputeLayer *hal = nullptr;

	size_t n_nodes() const { return ((size_t)2 << log_len) - 1; }
	std::vector<Digest> inner_nodes() const
	{
		std::vector<Digest> out(n_nodes());
		hal->copy_d2h(nodes, reinterpret_cast<B128 *>(out.data()), nodes.len());
		return out;
	}
	Digest root() const { return gather({n_nodes() - 1})[0]; }
	// binary_merkle_tree.rs:109-116
	std::vector<Digest> layer(size_t layer_depth) const
	{
		if (layer_depth > log_len) throw MerkleError("IncorrectLayerDepth");
		const size_t start = n_nodes() + 1 - ((size_t)1 << (layer_depth + 1)), n = (size_t)1 << layer_depth;
		std::vector<Digest> out(n);
		hal->copy_d2h(ComputeMemory::slice(nodes, 2 * start, 2 * (start + n)), reinterpret_cast<B128 *>(out.data()), 2 * n);
		return out;
	}
	// binary_merkle_tree.rs:121-141
	std::vector<Digest> branch(size_t index, size_t layer_depth) const
	{
		if (index >= ((size_t)1 << log_len) || layer_depth > log_len)
			throw MerkleError("IndexOutOfRange { max: " + std::to_string(((size_t)1 << log_len) - 1) + " }");
		std::vector<size_t> ids;
		for (size_t j = 0; j < log_len - layer_depth; j++)
			ids.push_back(((((size_t)1 << j) - 1) << (log_len + 1 - j)) | ((index >> j) ^ 1));
		return gather(ids);
	}

private:
	std::vector<Digest> gather(const std::vector<size_t> &node_ids) const
	{
		std::vector<uint64_t> offs;
		for (size_t id : node_ids) offs.push_back(2 * (uint64_t)id);
		std::vector<Digest> out(node_ids.size());
		static_assert(sizeof(Digest) == 2 * sizeof(B128), "a digest is two field elements wide");
		check(bn_gather_d2h(hal->raw_ctx(), nodes.ptr, offs.data(), offs.size(), 2, reinterpret_cast<bn_f128 *>(out.data())));
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
		tree.nodes = ComputeMemory::as_const(nodes);
		tree.hal = &hal_;
		return {Commitment{tree.root(), tree.log_len}, tree};
	}
	std::vector<Digest> layer(const BinaryMerkleTree &committed, size_t depth) const { return committed.layer(depth); }
	// prover.rs:73-83: the branch that is written to the transcript
	std::vector<Digest> prove_opening(const BinaryMerkleTree &committed, size_t layer_depth, size_t index) const
	{
		return committed.branch(index, layer_depth);
	}

private:
	ComputeLayer &hal_;
};

} // namespace binius_amd
