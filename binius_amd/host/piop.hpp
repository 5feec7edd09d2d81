// binius_amd/host/piop.hpp -- C++ mirror of the reference's PCS prover loop, the caller that decides the call shape the backend
// sees:
//
//   SumcheckBatchProver (front-loaded)     crates/core/src/protocols/sumcheck/prove/front_loaded.rs:33-203
//   CommitMeta                             crates/core/src/piop/verify.rs:30-98
//   PIOPSumcheckClaim                      crates/core/src/piop/verify.rs (n_vars, committed, transparent, sum)
//   make_sumcheck_claim_descs              crates/core/src/piop/verify.rs:192-270
//   merge_multilins / commit               crates/core/src/piop/prove.rs:66-146
//   prove                                  crates/core/src/piop/prove.rs:148-303
//   prove_interleaved_fri_sumcheck         crates/core/src/piop/prove.rs:306-395
//
// over the mirrors of BivariateSumcheckProver (sumcheck.hpp) and FRIFolder / commit_interleaved (fri.hpp).  Protocol bookkeeping
// only; every hypercube-sized operation is a HAL call.  The Fiat-Shamir transcript is replaced by what it would have produced
// (batch coefficients, one challenge per round) and by a record of what the prover writes to it, in order -- the same
// convention as the other mirrors (compute_test_utils' tests sample from a transcript; ours take a seeded stream).
// F = P = BinaryField128b (the device field; packed committed multilinears are already its elements).
#pragma once
#include <chrono>
#include <deque>
#include <memory>

#include "fri.hpp"
#include "sumcheck.hpp"

namespace binius_amd {

class PiopError : public Error {
public:
	explicit PiopError(const std::string &what) : Error(InputValidation, what) {}
};

// RoundCoeffs arithmetic of the batch prover (protocols/sumcheck/common.rs:108-150)
inline void round_coeffs_add_scaled(std::vector<B128> &acc, const std::vector<B128> &rhs, B128 scale)
{
	if (acc.size() < rhs.size()) acc.resize(rhs.size(), B128::ZERO());
	for (size_t i = 0; i < rhs.size(); i++) acc[i] = acc[i] + rhs[i] * scale;
}

// What the prover writes to the transcript, in order (ProverTranscript::message()): scalars and digests
struct PiopTranscript {
	struct Item {
		enum Kind { RoundProof, MultilinearEvals, FriCommitment, FriTerminate } kind;
		std::vector<B128> scalars;
		Digest digest{};
	};
	std::vector<Item> items;
	void write_scalar_slice(Item::Kind k, const std::vector<B128> &v) { items.push_back(Item{k, v, Digest{}}); }
	void write_digest(const Digest &d) { items.push_back(Item{Item::FriCommitment, {}, d}); }
};

// front_loaded.rs:33-203 with the batch coefficients handed in (new_prebatched, :79-107)
class SumcheckBatchProver {
public:
	SumcheckBatchProver(std::vector<std::unique_ptr<BivariateSumcheckProver>> provers, const std::vector<B128> &batch_coeffs)
	{
		for (size_t i = 1; i < provers.size(); i++)
			if (provers[i]->n_vars() < provers[i - 1]->n_vars()) throw SumcheckError("ClaimsOutOfOrder");
		if (batch_coeffs.size() != provers.size()) throw SumcheckError("IncorrectNumberOfBatchCoeffs");
		for (size_t i = 0; i < provers.size(); i++) provers_.emplace_back(std::move(provers[i]), batch_coeffs[i]);
	}
	size_t total_rounds() const { return provers_.empty() ? 0 : provers_.back().first->n_vars(); }
	size_t n_live() const { return provers_.size(); }

	// :123-140
	void send_round_proof(PiopTranscript &transcript)
	{
		finish_claim_provers(transcript);
		std::vector<B128> round_coeffs;
		for (auto &pc : provers_) {
			const std::vector<B128> prover_coeffs = pc.first->execute(pc.second);
			round_coeffs_add_scaled(round_coeffs, prover_coeffs, pc.second);
		}
		if (!round_coeffs.empty()) round_coeffs.pop_back(); // RoundCoeffs::truncate (common.rs:101-105)
		transcript.write_scalar_slice(PiopTranscript::Item::RoundProof, round_coeffs);
	}
	// :143-158
	void receive_challenge(B128 challenge)
	{
		for (auto &pc : provers_) pc.first->fold(challenge);
		round_++;
	}
	// :161-172
	std::vector<std::vector<B128>> finish(PiopTranscript &transcript)
	{
		finish_claim_provers(transcript);
		if (!provers_.empty()) throw SumcheckError("ExpectedFold");
		return multilinear_evals_;
	}

private:
	// :109-120
	void finish_claim_provers(PiopTranscript &transcript)
	{
		while (!provers_.empty() && provers_.front().first->n_vars() == round_) {
			std::vector<B128> evals = provers_.front().first->finish();
			provers_.pop_front();
			transcript.write_scalar_slice(PiopTranscript::Item::MultilinearEvals, evals);
			multilinear_evals_.push_back(std::move(evals));
		}
	}
	std::deque<std::pair<std::unique_ptr<BivariateSumcheckProver>, B128>> provers_;
	std::vector<std::vector<B128>> multilinear_evals_;
	size_t round_ = 0;
};

// piop/verify.rs:30-98
class CommitMeta {
public:
	explicit CommitMeta(std::vector<size_t> n_multilins_by_vars) : n_multilins_by_vars_(std::move(n_multilins_by_vars))
	{
		size_t total_elems = 0;
		for (size_t n_vars = 0; n_vars < n_multilins_by_vars_.size(); n_vars++) {
			offsets_by_vars_.push_back(total_multilins_);
			total_multilins_ += n_multilins_by_vars_[n_vars];
			total_elems += n_multilins_by_vars_[n_vars] << n_vars;
		}
		while (((size_t)1 << total_vars_) < total_elems) total_vars_++; // next_power_of_two().ilog2()
	}
	static CommitMeta with_vars(const std::vector<size_t> &n_varss)
	{
		std::vector<size_t> by_vars;
		for (size_t v : n_varss) {
			if (by_vars.size() <= v) by_vars.resize(v + 1, 0);
			by_vars[v]++;
		}
		return CommitMeta(by_vars);
	}
	size_t total_vars() const { return total_vars_; }
	size_t total_multilins() const { return total_multilins_; }
	size_t max_n_vars() const { return n_multilins_by_vars_.empty() ? 0 : n_multilins_by_vars_.size() - 1; }
	const std::vector<size_t> &n_multilins_by_vars() const { return n_multilins_by_vars_; }
	std::pair<size_t, size_t> range_by_vars(size_t n_vars) const { return {offsets_by_vars_[n_vars], offsets_by_vars_[n_vars] + n_multilins_by_vars_[n_vars]}; }

private:
	std::vector<size_t> n_multilins_by_vars_, offsets_by_vars_;
	size_t total_vars_ = 0, total_multilins_ = 0;
};

struct PIOPSumcheckClaim {
	size_t n_vars, committed, transparent;
	B128 sum;
};

// piop/verify.rs:176-270: per number of variables, the ranges of committed / transparent multilinears and the product claims over
// their concatenation (committed first)
struct SumcheckClaimDesc {
	size_t committed_begin = 0, committed_end = 0, transparent_begin = 0, transparent_end = 0;
	std::vector<IndexCompositionBivariate> compositions;
	std::vector<B128> sums;
	size_t n_committed() const { return committed_end - committed_begin; }
	size_t n_transparent() const { return transparent_end - transparent_begin; }
};
inline std::vector<SumcheckClaimDesc> make_sumcheck_claim_descs(const CommitMeta &commit_meta, const std::vector<size_t> &transparent_n_vars,
                                                                const std::vector<PIOPSumcheckClaim> &claims)
{
	std::vector<SumcheckClaimDesc> descs(commit_meta.max_n_vars() + 1);
	size_t last_offset = 0;
	for (size_t v = 0; v < descs.size(); v++) {
		descs[v].committed_begin = last_offset;
		last_offset += commit_meta.n_multilins_by_vars()[v];
		descs[v].committed_end = last_offset;
	}
	size_t current_n_vars = 0;
	for (size_t tv : transparent_n_vars) {
		if (tv < current_n_vars) throw PiopError("TransparentsNotSorted");
		if (tv > current_n_vars) {
			const size_t offset = descs[current_n_vars].transparent_end;
			current_n_vars = tv;
			if (current_n_vars >= descs.size()) throw PiopError("SumcheckClaimVariablesMismatch");
			descs[current_n_vars].transparent_begin = descs[current_n_vars].transparent_end = offset;
		}
		descs[current_n_vars].transparent_end++;
	}
	for (size_t i = 0; i < claims.size(); i++) {
		const PIOPSumcheckClaim &c = claims[i];
		if (c.n_vars >= descs.size()) throw PiopError("SumcheckClaimVariablesMismatch { index: " + std::to_string(i) + " }");
		SumcheckClaimDesc &d = descs[c.n_vars];
		if (c.committed < d.committed_begin || c.committed >= d.committed_end || c.transparent < d.transparent_begin || c.transparent >= d.transparent_end)
			throw PiopError("SumcheckClaimVariablesMismatch { index: " + std::to_string(i) + " }");
		const size_t n_ml = d.n_committed() + d.n_transparent();
		d.compositions.push_back(IndexCompositionBivariate{n_ml, {c.committed - d.committed_begin, d.n_committed() + c.transparent - d.transparent_begin}});
		d.sums.push_back(c.sum);
	}
	return descs;
}

// merge_multilins (piop/prove.rs:66-104) for P = F (LOG_WIDTH = 0): the multilinears in REVERSE order, each with bit-reversed
// indices, then zeros up to 2^total_vars.  Host arithmetic on host data, as in the reference (commit runs before any HAL exists).
inline std::vector<B128> merge_multilins(const std::vector<std::vector<B128>> &multilins, size_t total_vars)
{
	std::vector<B128> message((size_t)1 << total_vars, B128::ZERO());
	size_t at = 0;
	for (size_t r = multilins.size(); r-- > 0;) {
		const std::vector<B128> &evals = multilins[r];
		size_t log_len = 0;
		while (((size_t)1 << log_len) < evals.size()) log_len++;
		if (((size_t)1 << log_len) != evals.size() || at + evals.size() > message.size()) throw PiopError("merge_multilins: lengths");
		for (size_t i = 0; i < evals.size(); i++) {
			size_t rev = 0;
			for (size_t b = 0; b < log_len; b++) rev |= ((i >> b) & 1) << (log_len - 1 - b);
			message[at + rev] = evals[i];
		}
		at += evals.size();
	}
	return message;
}

struct PiopProveOutput {
	PiopTranscript transcript;
	std::vector<std::vector<B128>> multilinear_evals; // per prover, in finishing order
	std::vector<B128> terminate_codeword;
	size_t n_provers = 0;
	uint64_t phase_ns[3] = {0, 0, 0}; // BNH_PROF=1: send_round_proof, receive_challenge, execute_fold_round over all rounds
};

// prove_interleaved_fri_sumcheck (piop/prove.rs:306-395); `challenges[round]` is what transcript.sample() would have returned
inline PiopProveOutput prove_interleaved_fri_sumcheck(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, size_t n_rounds, const FRIParams &fri_params,
                                                      const AdditiveNTT &ntt, BinaryMerkleTreeProver &merkle_prover,
                                                      std::vector<std::unique_ptr<BivariateSumcheckProver>> sumcheck_provers,
                                                      const std::vector<B128> &batch_coeffs, FSlice codeword, const BinaryMerkleTree &committed,
                                                      const std::vector<B128> &challenges)
{
	if (challenges.size() < n_rounds) throw PiopError("not enough challenges for the rounds of the protocol");
	PiopProveOutput out;
	out.n_provers = sumcheck_provers.size();
	FRIFolder fri_prover(hal, fri_params, ntt, merkle_prover, codeword, committed);
	SumcheckBatchProver sumcheck_batch_prover(std::move(sumcheck_provers), batch_coeffs);
	const bool prof = AbiProf::on(); // BNH_PROF=1: wall time of the three steps of a round, summed (diagnostic)
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ns = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
		return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
	};
	for (size_t round = 0; round < n_rounds; round++) {
		const auto t0 = prof ? now() : std::chrono::steady_clock::time_point{};
		sumcheck_batch_prover.send_round_proof(out.transcript);
		const auto t1 = prof ? now() : t0;
		const B128 challenge = challenges[round];
		sumcheck_batch_prover.receive_challenge(challenge);
		const auto t2 = prof ? now() : t0;
		auto [has_commitment, round_commitment] = fri_prover.execute_fold_round(dev_alloc, challenge);
		if (has_commitment) out.transcript.write_digest(round_commitment);
		if (prof) {
			out.phase_ns[0] += ns(t0, t1);
			out.phase_ns[1] += ns(t1, t2);
			out.phase_ns[2] += ns(t2, now());
		}
	}
	out.multilinear_evals = sumcheck_batch_prover.finish(out.transcript);
	// fri_prover.finish_proof (fri/prove.rs:484-520): the terminate codeword goes to the transcript; the query phase (index sampling,
	// openings) is the query prover's and is covered by tests/test_gpu_fri.py
	auto fin = fri_prover.finalize();
	out.terminate_codeword = fin.first;
	out.transcript.write_scalar_slice(PiopTranscript::Item::FriTerminate, out.terminate_codeword);
	return out;
}

// prove (piop/prove.rs:148-303) from the point where the committed multilinears are on the device: the sumcheck claim descriptions,
// one BivariateSumcheckProver per number of variables with at least one committed multilinear (claims or not: unconstrained
// columns still owe their final evaluations), then the interleaved loop.
//   committed       device slices of the packed committed multilinears, ascending by number of variables (commit order)
//   transparents    device slices, ascending by number of variables
//   batch_coeffs    one per prover (transcript.sample_vec(provers.len()), front_loaded.rs:63-68)
inline PiopProveOutput piop_prove(ComputeLayer &hal, DeviceBumpAllocator &dev_alloc, HostBumpAllocator &host_alloc, const FRIParams &fri_params,
                                  const AdditiveNTT &ntt, BinaryMerkleTreeProver &merkle_prover, const CommitMeta &commit_meta, const BinaryMerkleTree &committed,
                                  FSlice codeword, const std::vector<FSlice> &committed_multilins, const std::vector<FSlice> &transparent_multilins,
                                  const std::vector<PIOPSumcheckClaim> &claims, const std::vector<B128> &batch_coeffs, const std::vector<B128> &challenges)
{
	if (committed_multilins.size() != commit_meta.total_multilins()) throw PiopError("committed multilinears do not match the commit metadata");
	std::vector<size_t> transparent_n_vars;
	for (const FSlice &t : transparent_multilins) {
		size_t l = 0;
		while (((size_t)1 << l) < t.len()) l++;
		if (((size_t)1 << l) != t.len()) throw PiopError("transparent multilinear length is not a power of two");
		transparent_n_vars.push_back(l);
	}
	const std::vector<SumcheckClaimDesc> descs = make_sumcheck_claim_descs(commit_meta, transparent_n_vars, claims);
	std::vector<std::unique_ptr<BivariateSumcheckProver>> provers;
	for (size_t n_vars = 0; n_vars < descs.size(); n_vars++) {
		const SumcheckClaimDesc &d = descs[n_vars];
		if (d.n_committed() == 0) continue; // (prove.rs:262-268)
		std::vector<FSlice> multilins;
		for (size_t i = d.committed_begin; i < d.committed_end; i++) multilins.push_back(committed_multilins[i]);
		for (size_t i = d.transparent_begin; i < d.transparent_end; i++) multilins.push_back(transparent_multilins[i]);
		provers.push_back(std::make_unique<BivariateSumcheckProver>(hal, dev_alloc, host_alloc, n_vars, d.compositions, d.sums, multilins));
	}
	if (batch_coeffs.size() != provers.size()) throw PiopError("one batch coefficient per sumcheck prover: " + std::to_string(provers.size()) + " expected");
	return prove_interleaved_fri_sumcheck(hal, dev_alloc, commit_meta.total_vars(), fri_params, ntt, merkle_prover, std::move(provers), batch_coeffs, codeword,
	                                      committed, challenges);
}

} // namespace binius_amd
