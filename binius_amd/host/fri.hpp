// binius_amd/host/fri.hpp -- C++ mirror of the reference's FRI commit and fold phases with the
// codewords and Merkle trees resident on the device.
//
// Mirrors
//   FRIParams                              crates/core/src/protocols/fri/common.rs:25-171
//   ReedSolomonCode::encode_ext_batch_inplace
//                                          crates/core/src/reed_solomon/reed_solomon.rs:104-184
//   commit_interleaved(_with)              crates/core/src/protocols/fri/prove.rs:88-198
//   FRIFolder (execute_fold_round, finalize)
//                                          crates/core/src/protocols/fri/prove.rs:219-482
//   FRIQueryProver (prove_query, vcs_optimal_layers), prove_coset_opening
//                                          crates/core/src/protocols/fri/prove.rs:523-661
// with F = BinaryField128b, FA = BinaryField32b (crates/core/src/constraint_system/common.rs:22).
// The reference keeps a host copy of every codeword because its Merkle prover hashes host slices
// (prove.rs:396-399); here nothing but roots, layers, branches and opened cosets is read back.
#pragma once
#include <numeric>

#include "merkle.hpp"

namespace binius_amd {

class FriError : public Error {
public:
	explicit FriError(const std::string &what) : Error(InputValidation, what) {}
};

// common.rs:25-171
class FRIParams {
public:
	FRIParams(size_t log_dim, size_t log_inv_rate, size_t log_batch_size, std::vector<size_t> fold_arities, size_t n_test_queries)
	    : log_dim_(log_dim), log_inv_rate_(log_inv_rate), log_batch_size_(log_batch_size), fold_arities_(std::move(fold_arities)),
	      n_test_queries_(n_test_queries)
	{
		if (std::accumulate(fold_arities_.begin(), fold_arities_.end(), (size_t)0) >= log_dim + log_batch_size)
			throw FriError("InvalidFoldAritySequence"); // (:47-49)
	}
	size_t log_dim() const { return log_dim_; }
	size_t log_inv_rate() const { return log_inv_rate_; }
	size_t log_batch_size() const { return log_batch_size_; }
	size_t n_test_queries() const { return n_test_queries_; }
	const std::vector<size_t> &fold_arities() const { return fold_arities_; }
	size_t rs_log_len() const { return log_dim_ + log_inv_rate_; }
	size_t n_fold_rounds() const { return log_dim_ + log_batch_size_; }
	size_t n_oracles() const { return fold_arities_.size(); }
	size_t log_len() const { return rs_log_len() + log_batch_size_; }
	size_t index_bits() const { return fold_arities_.empty() ? 0 : log_len() - fold_arities_[0]; }
	size_t n_final_challenges() const { return n_fold_rounds() - std::accumulate(fold_arities_.begin(), fold_arities_.end(), (size_t)0); }
	// vcs_optimal_layers_depths_iter (common.rs:174-190) with optimal_verify_layer (merkle_tree/scheme.rs:47-49)
	std::vector<size_t> optimal_layer_depths() const
	{
		std::vector<size_t> out;
		size_t log_n_cosets = log_len(), cap = 0;
		while (((size_t)1 << cap) < n_test_queries_) cap++;
		for (size_t arity : fold_arities_) {
			log_n_cosets -= arity;
			out.push_back(std::min(cap, log_n_cosets));
		}
		return out;
	}

private:
	size_t log_dim_, log_inv_rate_, log_batch_size_;
	std::vector<size_t> fold_arities_;
	size_t n_test_queries_;
};

// reed_solomon.rs:104-184 on a device buffer whose first 2^(log_dim + log_batch_size) elements hold the
// interleaved message: repeat it 2^log_inv_rate times, then ONE batched NTT over the B32 columns
inline void encode_ext_batch_inplace(ComputeLayer &hal, const AdditiveNTT &ntt, const FRIParams &p, FSliceMut code, size_t log_batch_size)
{
	constexpr size_t kFaLevel = 5, kLogDegree = 2; // BinaryField32b; BinaryField128b over it
	if (p.rs_log_len() > ntt.log_domain_size() || ntt.tower_level() != kFaLevel) throw FriError("EncoderSubspaceMismatch");
	const size_t want = (size_t)1 << (p.rs_log_len() + log_batch_size);
	if (code.len() != want)
		throw FriError("IncorrectBufferLength { expected: " + std::to_string(want) + ", actual: " + std::to_string(code.len()) + " }");
	const size_t msg_len = (size_t)1 << (p.log_dim() + log_batch_size);
	const FSlice first = ComputeMemory::slice(ComputeMemory::as_const(code), 0, msg_len);
	for (size_t j = 1; j < ((size_t)1 << p.log_inv_rate()); j++) {
		FSliceMut dst = ComputeMemory::slice_mut(code, j * msg_len, (j + 1) * msg_len);
		hal.copy_d2d(first, dst);
	}
	ntt.forward_transform(code.ptr, kFaLevel, NTTShape{log_batch_size + kLogDegree, p.rs_log_len(), 0}, 0, 0, p.log_inv_rate());
}

struct CommitOutput { // prove.rs:70-74
	Digest commitment;
	BinaryMerkleTree committed;
	FSliceMut codeword;
};

// prove.rs:88-198; `message`: device slice of 2^(log_dim + log_batch_size) elements
inline CommitOutput commit_interleaved(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, const FRIParams &p, const AdditiveNTT &ntt,
                                       BinaryMerkleTreeProver &merkle_prover, FSlice message)
{
	const size_t log_elems = p.log_dim() + p.log_batch_size();
	if (message.len() != (size_t)1 << log_elems) throw FriError("InvalidArgs(interleaved message length does not match code parameters)");
	FSliceMut encoded = dev_alloc.alloc((size_t)1 << (log_elems + p.log_inv_rate()));
	FSliceMut head = ComputeMemory::slice_mut(encoded, 0, message.len());
	hal.copy_d2d(message, head);
	encode_ext_batch_inplace(hal, ntt, p, encoded, p.log_batch_size());
	const size_t coset_log_len = p.fold_arities().empty() ? log_elems : p.fold_arities()[0];
	auto [commitment, tree] = merkle_prover.commit(ComputeMemory::as_const(encoded), (size_t)1 << coset_log_len, dev_alloc);
	return CommitOutput{commitment.root, tree, encoded};
}

struct CosetOpening { // what prove_coset_opening writes to the transcript (prove.rs:631-661)
	std::vector<B128> values;
	std::vector<Digest> branch;
};

// prove.rs:523-628
class FRIQueryProver {
public:
	FRIQueryProver(ComputeLayer &hal, const FRIParams &p, BinaryMerkleTreeProver &merkle_prover, FSlice codeword, BinaryMerkleTree committed,
	               std::vector<std::pair<FSlice, BinaryMerkleTree>> round_committed)
	    : hal_(hal), p_(p), merkle_prover_(merkle_prover), codeword_(codeword), committed_(committed), round_committed_(std::move(round_committed))
	{
	}
	size_t n_oracles() const { return p_.n_oracles(); }
	std::vector<std::vector<Digest>> vcs_optimal_layers() const
	{
		std::vector<std::vector<Digest>> out;
		const auto depths = p_.optimal_layer_depths();
		for (size_t i = 0; i < depths.size(); i++)
			out.push_back(merkle_prover_.layer(i == 0 ? committed_ : round_committed_[i - 1].second, depths[i]));
		return out;
	}
	std::vector<CosetOpening> prove_query(size_t index) const
	{
		std::vector<CosetOpening> out;
		const auto &arities = p_.fold_arities();
		if (arities.empty()) return out;
		const auto depths = p_.optimal_layer_depths();
		out.push_back(open(codeword_, committed_, index, arities[0], depths[0]));
		// (the last committed oracle has no arity after it and is not opened: izip stops at the shorter side)
		for (size_t i = 0; i + 1 < arities.size() && i < round_committed_.size(); i++) {
			index >>= arities[i + 1];
			out.push_back(open(round_committed_[i].first, round_committed_[i].second, index, arities[i + 1], depths[i + 1]));
		}
		return out;
	}

private:
	CosetOpening open(FSlice codeword, const BinaryMerkleTree &tree, size_t coset_index, size_t log_coset_size, size_t layer_depth) const
	{
		CosetOpening o;
		o.values.resize((size_t)1 << log_coset_size);
		const size_t lo = coset_index << log_coset_size;
		hal_.copy_d2h(ComputeMemory::slice(codeword, lo, lo + o.values.size()), o.values);
		o.branch = merkle_prover_.prove_opening(tree, layer_depth, coset_index);
		return o;
	}
	ComputeLayer &hal_;
	const FRIParams &p_;
	BinaryMerkleTreeProver &merkle_prover_;
	FSlice codeword_;
	BinaryMerkleTree committed_;
	std::vector<std::pair<FSlice, BinaryMerkleTree>> round_committed_;
};

// prove.rs:219-482
class FRIFolder {
public:
	FRIFolder(ComputeLayer &hal, const FRIParams &p, const AdditiveNTT &ntt, BinaryMerkleTreeProver &merkle_prover, FSlice codeword,
	          BinaryMerkleTree committed)
	    : hal_(hal), p_(p), ntt_(ntt), merkle_prover_(merkle_prover), codeword_(codeword), committed_(committed)
	{
		if (codeword.len() < (size_t)1 << p.log_len())
			throw FriError("InvalidArgs(Reed-Solomon code length must match interleaved codeword length)");
		if (!p.fold_arities().empty()) next_commit_round_ = (long)p.fold_arities()[0];
	}
	size_t n_rounds() const { return p_.n_fold_rounds(); }
	size_t curr_round() const { return curr_round_; }
	size_t current_codeword_len() const { return round_committed_.empty() ? codeword_.len() : round_committed_.back().first.len(); }
	const std::vector<std::pair<FSlice, BinaryMerkleTree>> &round_committed() const { return round_committed_; }

	// FoldRoundOutput: {false, _} = NoCommitment, {true, root} = Commitment(root)
	std::pair<bool, Digest> execute_fold_round(DeviceBumpAllocator &allocator, B128 challenge)
	{
		unprocessed_.push_back(challenge);
		curr_round_++;
		if (next_commit_round_ < 0 || (size_t)next_commit_round_ != curr_round_) return {false, Digest{}};
		const size_t n_ch = unprocessed_.size();
		FSliceMut folded{};
		if (!round_committed_.empty()) {
			const FSlice prev = round_committed_.back().first;
			size_t log_prev = 0;
			while (((size_t)1 << log_prev) < prev.len()) log_prev++;
			folded = allocator.alloc(prev.len() >> n_ch);
			hal_.execute([&](ComputeLayerExecutor &exec) {
				exec.fri_fold(ntt_, log_prev, 0, unprocessed_, prev, folded);
				return std::vector<B128>{};
			});
		} else {
			folded = allocator.alloc((size_t)1 << (p_.rs_log_len() - (n_ch - p_.log_batch_size())));
			hal_.execute([&](ComputeLayerExecutor &exec) {
				exec.fri_fold(ntt_, p_.rs_log_len(), p_.log_batch_size(), unprocessed_, codeword_, folded);
				return std::vector<B128>{};
			});
		}
		unprocessed_.clear();
		// the next arity as the coset size, or the final challenges when no oracle follows (:401-407)
		const size_t k = round_committed_.size() + 1;
		const size_t coset_size = (size_t)1 << (k < p_.fold_arities().size() ? p_.fold_arities()[k] : p_.n_final_challenges());
		auto [commitment, tree] = merkle_prover_.commit(ComputeMemory::as_const(folded), coset_size, allocator);
		round_committed_.push_back({ComputeMemory::as_const(folded), tree});
		const size_t n = round_committed_.size();
		next_commit_round_ = n < p_.fold_arities().size() ? next_commit_round_ + (long)p_.fold_arities()[n] : -1;
		return {true, commitment.root};
	}

	// (terminate_codeword on the host, query prover)  (prove.rs:444-482)
	std::pair<std::vector<B128>, FRIQueryProver> finalize()
	{
		if (curr_round_ != n_rounds()) throw FriError("EarlyProverFinish");
		const FSlice last = round_committed_.empty() ? codeword_ : round_committed_.back().first;
		std::vector<B128> terminate(last.len());
		hal_.copy_d2h(last, terminate);
		return {std::move(terminate), FRIQueryProver(hal_, p_, merkle_prover_, codeword_, committed_, round_committed_)};
	}

private:
	ComputeLayer &hal_;
	const FRIParams &p_;
	const AdditiveNTT &ntt_;
	BinaryMerkleTreeProver &merkle_prover_;
	FSlice codeword_;
	BinaryMerkleTree committed_;
	std::vector<std::pair<FSlice, BinaryMerkleTree>> round_committed_;
	size_t curr_round_ = 0;
	long next_commit_round_ = -1;
	std::vector<B128> unprocessed_;
};

} // namespace binius_amd
