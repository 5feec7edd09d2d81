"""Device-resident mirror of the reference's FRI commit and fold phases (SURVEY.md section 8(f) items
1 and 2): Reed-Solomon encoding by the batched additive NTT, the Groestl-256 Merkle commitment of the
interleaved codeword, and the fold rounds with a commitment per oracle -- all on the GPU; only
digests, the terminal codeword and the queried cosets come back to the host.

Mirrors
  FRIParams                       crates/core/src/protocols/fri/common.rs:25-171
  ReedSolomonCode.encode_ext_batch_inplace
                                  crates/core/src/reed_solomon/reed_solomon.rs:104-184
  commit_interleaved(_with)       crates/core/src/protocols/fri/prove.rs:88-198
  FRIFolder                       crates/core/src/protocols/fri/prove.rs:219-520
  FRIQueryProver, prove_coset_opening
                                  crates/core/src/protocols/fri/prove.rs:523-661
with F = BinaryField128b, FA = BinaryField32b (crates/core/src/constraint_system/common.rs:22) and
the BinaryMerkleTreeProver<Groestl256, Groestl256ByteCompression> of binius_amd/merkle.py.

Where the reference keeps a host copy of every codeword (prove.rs:396-399, needed there because the
Merkle tree is built on the host), this mirror keeps the device slice and reads back only the cosets
a query opens.
"""
import numpy as np

from ._ffi import BN_ERR_INPUT_VALIDATION, BnError, ntt_s_evals
from .merkle import BinaryMerkleTreeProver

FA_LEVEL = 5  # BinaryField32b
LOG_DEGREE = 2  # BinaryField128b over BinaryField32b


class FriError(BnError):
    pass


def _log2_ceil(n):
    return max(0, (n - 1).bit_length())


class FRIParams:
    """common.rs:25-171."""

    def __init__(self, log_dim, log_inv_rate, log_batch_size, fold_arities, n_test_queries):
        fold_arities = list(fold_arities)
        # common.rs:47-54
        if sum(fold_arities) >= log_dim + log_batch_size:
            raise FriError(BN_ERR_INPUT_VALIDATION, "InvalidFoldAritySequence")
        self.log_dim, self.log_inv_rate, self.log_batch_size = log_dim, log_inv_rate, log_batch_size
        self.fold_arities, self.n_test_queries = fold_arities, n_test_queries

    def rs_log_len(self):
        return self.log_dim + self.log_inv_rate

    def n_fold_rounds(self):
        return self.log_dim + self.log_batch_size

    def n_oracles(self):
        return len(self.fold_arities)

    def index_bits(self):
        return self.log_len() - self.fold_arities[0] if self.fold_arities else 0

    def n_final_challenges(self):
        return self.n_fold_rounds() - sum(self.fold_arities)

    def log_len(self):
        return self.rs_log_len() + self.log_batch_size

    def optimal_layer_depths(self):
        """vcs_optimal_layers_depths_iter (common.rs:174-190) with scheme.rs:47-49."""
        out, log_n_cosets = [], self.log_len()
        for arity in self.fold_arities:
            log_n_cosets -= arity
            out.append(min(_log2_ceil(self.n_test_queries), log_n_cosets))
        return out


class AdditiveNTT:
    """The on-the-fly twiddle basis of SingleThreadedNTT::<B32>::new(log_domain) (twiddle.rs:93-124)."""

    def __init__(self, log_domain, tw_level=FA_LEVEL):
        self.log_domain, self.tw_level = log_domain, tw_level
        self.s_evals = ntt_s_evals(tw_level, log_domain)


def encode_ext_batch_inplace(hal, ntt, params, code, log_batch_size):
    """reed_solomon.rs:104-184 on a device buffer `code` of 2^(log_len + log_batch_size) elements whose
    first 2^(log_dim + log_batch_size) hold the interleaved message."""
    log_len = params.rs_log_len()
    if params.rs_log_len() > ntt.log_domain:
        raise FriError(BN_ERR_INPUT_VALIDATION, "EncoderSubspaceMismatch")
    if code.len != 1 << (log_len + log_batch_size):
        raise FriError(BN_ERR_INPUT_VALIDATION, "IncorrectBufferLength { expected: %d, actual: %d }" % (1 << (log_len + log_batch_size), code.len))
    msg_len = 1 << (params.log_dim + log_batch_size)
    first = code.slice(0, msg_len)
    for j in range(1, 1 << params.log_inv_rate):  # repeat the message to fill the buffer (:146-153)
        hal.copy_d2d(first, code.slice(j * msg_len, (j + 1) * msg_len))
    # NTTShape{log_x: log_batch_size + LOG_DEGREE, log_y: log_len}, coset 0, skip_rounds = log_inv_rate (:155-160)
    hal.ntt_forward(code.ptr, FA_LEVEL, ntt.tw_level, ntt.s_evals, ntt.log_domain, log_batch_size + LOG_DEGREE, log_len, 0, 0, 0, params.log_inv_rate)


class CommitOutput:
    def __init__(self, commitment, committed, codeword):
        self.commitment, self.committed, self.codeword = commitment, committed, codeword


def commit_interleaved(hal, dev_alloc, params, ntt, merkle_prover, message):
    """prove.rs:88-198.  `message`: device slice of 2^(log_dim + log_batch_size) elements."""
    log_elems = params.log_dim + params.log_batch_size
    if message.len != 1 << log_elems:
        raise FriError(BN_ERR_INPUT_VALIDATION, "InvalidArgs(interleaved message length does not match code parameters)")
    encoded = dev_alloc.alloc(1 << (log_elems + params.log_inv_rate))
    hal.copy_d2d(message, encoded.slice(0, message.len))
    encode_ext_batch_inplace(hal, ntt, params, encoded, params.log_batch_size)
    coset_log_len = params.fold_arities[0] if params.fold_arities else log_elems
    (root, _depth), tree = merkle_prover.commit(encoded, 1 << coset_log_len)
    return CommitOutput(root, tree, encoded)


class FRIFolder:
    """prove.rs:219-520."""

    def __init__(self, hal, params, ntt, merkle_prover, codeword, committed):
        if codeword.len < 1 << params.log_len():
            raise FriError(BN_ERR_INPUT_VALIDATION, "InvalidArgs(Reed-Solomon code length must match interleaved codeword length)")
        self.hal, self.params, self.ntt, self.merkle_prover = hal, params, ntt, merkle_prover
        self.codeword, self.codeword_committed = codeword, committed
        self.round_committed = []  # (device codeword, tree)
        self.curr_round = 0
        self.next_commit_round = params.fold_arities[0] if params.fold_arities else None
        self.unprocessed_challenges = []

    def n_rounds(self):
        return self.params.n_fold_rounds()

    def current_codeword_len(self):
        return self.round_committed[-1][0].len if self.round_committed else self.codeword.len

    def execute_fold_round(self, allocator, challenge):
        """Returns None (FoldRoundOutput::NoCommitment) or the new oracle's root."""
        p = self.params
        self.unprocessed_challenges.append(challenge)
        self.curr_round += 1
        if self.next_commit_round != self.curr_round:
            return None
        n_ch = len(self.unprocessed_challenges)
        if self.round_committed:
            prev = self.round_committed[-1][0]
            folded = allocator.alloc(prev.len >> n_ch)
            self.hal.fri_fold(self.ntt.s_evals, self.ntt.tw_level, self.ntt.log_domain, prev.len.bit_length() - 1, 0, self.unprocessed_challenges, prev, folded)
        else:
            folded = allocator.alloc(1 << (p.rs_log_len() - (n_ch - p.log_batch_size)))
            self.hal.fri_fold(self.ntt.s_evals, self.ntt.tw_level, self.ntt.log_domain, p.rs_log_len(), p.log_batch_size, self.unprocessed_challenges, self.codeword, folded)
        self.unprocessed_challenges = []
        # the next arity as the coset size, or the final challenges when no oracle follows (:401-407)
        k = len(self.round_committed) + 1
        coset_size = 1 << (p.fold_arities[k] if k < len(p.fold_arities) else p.n_final_challenges())
        (root, _depth), tree = self.merkle_prover.commit(folded, coset_size)
        self.round_committed.append((folded, tree))
        n = len(self.round_committed)
        self.next_commit_round = self.next_commit_round + p.fold_arities[n] if n < len(p.fold_arities) else None
        return root

    def finalize(self):
        """(terminate_codeword on the host, FRIQueryProver)  (prove.rs:444-482)."""
        if self.curr_round != self.n_rounds():
            raise FriError(BN_ERR_INPUT_VALIDATION, "EarlyProverFinish")
        last = self.round_committed[-1][0] if self.round_committed else self.codeword
        terminate = self.hal.copy_d2h(last)
        return terminate, FRIQueryProver(self.hal, self.params, self.merkle_prover, self.codeword, self.codeword_committed, self.round_committed)


class FRIQueryProver:
    """prove.rs:523-628."""

    def __init__(self, hal, params, merkle_prover, codeword, committed, round_committed):
        self.hal, self.params, self.merkle_prover = hal, params, merkle_prover
        self.codeword, self.codeword_committed, self.round_committed = codeword, committed, round_committed

    def n_oracles(self):
        return self.params.n_oracles()

    def vcs_optimal_layers(self):
        trees = [self.codeword_committed] + [t for _, t in self.round_committed]
        return [self.merkle_prover.layer(t, d) for t, d in zip(trees, self.params.optimal_layer_depths())]

    def _coset_opening(self, codeword, committed, coset_index, log_coset_size, layer_depth):
        """prove_coset_opening (:631-661): the coset's values, then the Merkle branch."""
        lo = coset_index << log_coset_size
        values = self.hal.copy_d2h(codeword.slice(lo, lo + (1 << log_coset_size)))
        return values, self.merkle_prover.prove_opening(committed, layer_depth, coset_index)

    def prove_query(self, index):
        p = self.params
        if not p.fold_arities:
            return []
        depths = p.optimal_layer_depths()
        out = [self._coset_opening(self.codeword, self.codeword_committed, index, p.fold_arities[0], depths[0])]
        # (izip stops at the shorter side: the last committed oracle has no arity after it and is not opened)
        for (codeword, committed), arity, depth in zip(self.round_committed, p.fold_arities[1:], depths[1:]):
            index >>= arity  # (:594-596: by the arity of THIS opening)
            out.append(self._coset_opening(codeword, committed, index, arity, depth))
        return out


__all__ = ["FRIParams", "AdditiveNTT", "encode_ext_batch_inplace", "commit_interleaved", "CommitOutput", "FRIFolder", "FRIQueryProver", "BinaryMerkleTreeProver"]
