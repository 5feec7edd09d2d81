"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's PCS prover loop, composed from the oracle's pinned pieces
(oracle/__init__.py: the scalar / PCLMULQDQ bivariate sumcheck prover, additive NTT, fri_fold, Groestl Merkle tree).  Only
tests/, bench-side checkers and tools' `--check` legs import this; the product path never does.

  batch_sumcheck_prove        SumcheckBatchProver::run            crates/core/src/protocols/sumcheck/prove/front_loaded.rs:33-203
  CommitMeta                  CommitMeta::{new, with_vars}        crates/core/src/piop/verify.rs:47-98
  make_sumcheck_claim_descs                                       crates/core/src/piop/verify.rs:192-270
  merge_multilins             (P = F: LOG_WIDTH = 0)              crates/core/src/piop/prove.rs:66-104
  piop_prove                  commit + prove + interleaved loop   crates/core/src/piop/prove.rs:106-395, fri/prove.rs:88-198, 309-420

Parity pin: the pieces are pinned in tests/test_oracle_*.py; this file adds bookkeeping only, and tests/test_oracle_piop.py checks
it against the verifier's own equations (batched round consistency, final evaluations = multilinear extensions at the reversed
challenges, the committed evaluations against the FRI terminate value through evaluate_piecewise_multilinear,
piop/verify.rs:343-358, 556-600)."""
import numpy as np

import oracle as o


def _prove_one(mls, n_vars, comps, sums, batch_coeff, challenges, threads, fast):
    """One BivariateSumcheckProver run to the end (v3/bivariate_product.rs:133-254): round coefficients [n_vars][3], finals."""
    if n_vars == 0:
        return [], [o.arr_to_ints(x)[0] for x in mls]
    if fast:
        got = o.fast_bivariate_sumcheck_prove(mls, n_vars, comps, sums, batch_coeff, challenges[:n_vars], threads=threads)
        if got is not None:
            return got
    return o.bivariate_sumcheck_prove(mls, n_vars, comps, sums, batch_coeff, challenges[:n_vars], threads=threads)


def batch_sumcheck_prove(provers, batch_coeffs, challenges, threads=1, fast=False):
    """front_loaded.rs:33-203.  provers: list of dict(n_vars, multilins (numpy arrays, OVERWRITTEN), comps, sums), ascending by
    n_vars.  The provers interact only through the shared challenges (all start in round 0), so each is run to the end by the
    pinned single-prover oracle and the batch prover's bookkeeping is restated on top: round r's proof is the sum over the
    provers still alive of batch_coeff_p * coeffs_p[r], truncated by its last coefficient (common.rs:101-105); a prover's final
    evaluations are written in the round that equals its number of variables (finish_claim_provers, :109-120).
    Returns (items, multilinear_evals): items = [("round_proof", [..]) | ("multilinear_evals", [..])] in transcript order."""
    assert all(provers[i]["n_vars"] <= provers[i + 1]["n_vars"] for i in range(len(provers) - 1)), "ClaimsOutOfOrder"
    assert len(batch_coeffs) == len(provers), "IncorrectNumberOfBatchCoeffs"
    runs = []
    for p, bc in zip(provers, batch_coeffs):
        coeffs, finals = _prove_one(p["multilins"], p["n_vars"], p["comps"], p["sums"], bc, challenges, threads, fast)
        if not p["comps"]:
            coeffs = [[] for _ in range(p["n_vars"])]  # (an empty composition list contributes a degree-0 polynomial, bivariate_product.rs:163-171)
        runs.append((coeffs, finals))
    total_rounds = provers[-1]["n_vars"] if provers else 0
    items, evals, first_live = [], [], 0

    def finish(round_):
        nonlocal first_live
        while first_live < len(provers) and provers[first_live]["n_vars"] == round_:
            items.append(("multilinear_evals", list(runs[first_live][1])))
            evals.append(list(runs[first_live][1]))
            first_live += 1

    for r in range(total_rounds):
        finish(r)
        acc = []
        for i in range(first_live, len(provers)):
            pc = runs[i][0][r]
            if len(acc) < len(pc):
                acc += [0] * (len(pc) - len(acc))
            for j, c in enumerate(pc):
                acc[j] ^= o.mul(c, batch_coeffs[i])
        if acc:
            acc.pop()
        items.append(("round_proof", acc))
    finish(total_rounds)
    assert first_live == len(provers)
    return items, evals


class CommitMeta:
    """piop/verify.rs:30-98"""

    def __init__(self, n_multilins_by_vars):
        self.n_multilins_by_vars = list(n_multilins_by_vars)
        self.offsets_by_vars, total, elems = [], 0, 0
        for n_vars, count in enumerate(self.n_multilins_by_vars):
            self.offsets_by_vars.append(total)
            total += count
            elems += count << n_vars
        self.total_multilins = total
        self.total_vars = max(0, (elems - 1).bit_length()) if elems else 0  # next_power_of_two().ilog2()

    @classmethod
    def with_vars(cls, n_varss):
        by = []
        for v in n_varss:
            if len(by) <= v:
                by += [0] * (v + 1 - len(by))
            by[v] += 1
        return cls(by)

    def max_n_vars(self):
        return max(0, len(self.n_multilins_by_vars) - 1)


def make_sumcheck_claim_descs(commit_meta, transparent_n_vars, claims):
    """piop/verify.rs:192-270.  claims: (n_vars, committed, transparent, sum).  Returns per n_vars a dict(committed=(b, e),
    transparent=(b, e), comps=[(i, j)], sums=[..]) with the compositions indexing committed ++ transparent of that size."""
    descs = [dict(committed=(0, 0), transparent=(0, 0), comps=[], sums=[]) for _ in range(commit_meta.max_n_vars() + 1)]
    last = 0
    for v, d in enumerate(descs):
        d["committed"] = (last, last + commit_meta.n_multilins_by_vars[v])
        last = d["committed"][1]
    cur = 0
    for tv in transparent_n_vars:
        assert tv >= cur, "TransparentsNotSorted"
        if tv > cur:
            off = descs[cur]["transparent"][1]
            cur = tv
            descs[cur]["transparent"] = (off, off)
        b, e = descs[cur]["transparent"]
        descs[cur]["transparent"] = (b, e + 1)
    for i, (n_vars, committed, transparent, s) in enumerate(claims):
        d = descs[n_vars]
        cb, ce = d["committed"]
        tb, te = d["transparent"]
        assert cb <= committed < ce and tb <= transparent < te, "SumcheckClaimVariablesMismatch { index: %d }" % i
        d["comps"].append((committed - cb, (ce - cb) + transparent - tb))
        d["sums"].append(s)
    return descs


def _bit_reverse_perm(log_len):
    idx = np.arange(1 << log_len, dtype=np.int64)
    rev = np.zeros_like(idx)
    for b in range(log_len):
        rev |= ((idx >> b) & 1) << (log_len - 1 - b)
    return rev


def merge_multilins(multilins, total_vars):
    """piop/prove.rs:66-104 for P = F: the multilinears in REVERSE order, each with bit-reversed indices, zero padded."""
    msg = np.zeros((1 << total_vars, 2), dtype=np.uint64)
    at = 0
    for evals in reversed(multilins):
        n = evals.shape[0]
        log_len = n.bit_length() - 1
        assert 1 << log_len == n and at + n <= msg.shape[0]
        chunk = np.empty_like(evals)
        chunk[_bit_reverse_perm(log_len)] = evals
        msg[at : at + n] = chunk
        at += n
    return msg


def fri_commit(p, s_evals, log_domain, message):
    """commit_interleaved (fri/prove.rs:88-198): repeat the message 2^log_inv_rate times, one batched NTT over the B32 columns,
    Merkle tree over cosets of 2^arity_0 (the whole message when there are no arities).  Returns (codeword, tree nodes)."""
    code = np.concatenate([message] * (1 << p.log_inv_rate))
    assert o.ntt_forward(code, 5, 5, s_evals, log_domain, p.log_batch_size + 2, p.rs_log_len(), 0, 0, 0, p.log_inv_rate) == 0
    coset_log_len = p.fold_arities[0] if p.fold_arities else p.log_dim + p.log_batch_size
    rc, nodes = o.merkle_build(code, 1 << coset_log_len)
    assert rc == 0
    return code, nodes


def piop_prove(committed, transparents, claims, p, batch_coeffs, challenges, threads=1, fast=False):
    """commit (piop/prove.rs:106-146) + prove (:148-303) + prove_interleaved_fri_sumcheck (:306-395) with F = P = B128.
    committed / transparents: lists of numpy arrays ascending by size (NOT modified); claims: (n_vars, committed, transparent,
    sum); p: an object with log_dim, log_inv_rate, log_batch_size, fold_arities and the FRIParams arithmetic
    (binius_amd._host.FRIParams has exactly that).  Returns (commitment root, transcript items, multilinear_evals,
    terminate codeword) with the items in writing order: round proofs and final evaluations as in batch_sumcheck_prove, a
    ("fri_commitment", 32 bytes) after the fold of every commit round, ("fri_terminate", [..]) last."""
    n_varss = [x.shape[0].bit_length() - 1 for x in committed]
    assert n_varss == sorted(n_varss), "CommittedsNotSorted"
    meta = CommitMeta.with_vars(n_varss)
    assert meta.total_vars == p.log_dim + p.log_batch_size
    log_domain = p.rs_log_len()
    s_evals = o.ntt_s_evals(5, log_domain)
    code, nodes = fri_commit(p, s_evals, log_domain, merge_multilins(committed, meta.total_vars))
    commitment = bytes(nodes[-1])
    descs = make_sumcheck_claim_descs(meta, [t.shape[0].bit_length() - 1 for t in transparents], claims)
    provers = []
    for n_vars, d in enumerate(descs):
        cb, ce = d["committed"]
        if ce == cb:
            continue  # (piop/prove.rs:262-268: sizes without a committed multilinear have no prover)
        tb, te = d["transparent"]
        mls = [committed[i].copy() for i in range(cb, ce)] + [transparents[i].copy() for i in range(tb, te)]
        provers.append(dict(n_vars=n_vars, multilins=mls, comps=d["comps"], sums=d["sums"]))
    assert len(batch_coeffs) == len(provers)
    n_rounds = meta.total_vars
    sc_items, evals = batch_sumcheck_prove(provers, batch_coeffs, challenges, threads=threads, fast=fast)
    # interleave: per round the round proof (preceded by the final evaluations of the provers that finish in it), then the FRI
    # commitment when the round ends an arity (FRIFolder::execute_fold_round, fri/prove.rs:309-420)
    per_round, tail, it = [], [], iter(sc_items)
    cur = []
    for item in it:
        cur.append(item)
        if item[0] == "round_proof":
            per_round.append(cur)
            cur = []
    tail = cur  # final evaluations written by finish()
    sc_rounds = len(per_round)
    items = []
    arities = list(p.fold_arities)
    next_commit = arities[0] if arities else None
    cur_code, cur_log_len, cur_log_batch, unprocessed, n_committed = code, p.rs_log_len(), p.log_batch_size, [], 0
    for r in range(n_rounds):
        if r < sc_rounds:
            items += per_round[r]
        else:
            # (the batch prover is empty from its last round on: it still writes an empty round proof every round)
            if r == sc_rounds:
                items += tail
                tail = []
            items.append(("round_proof", []))
        unprocessed.append(challenges[r])
        if next_commit is not None and r + 1 == next_commit:
            new_log_len = cur_log_len - (len(unprocessed) - cur_log_batch)
            nxt = o.arr(1 << new_log_len)
            assert o.fri_fold(s_evals, 5, log_domain, cur_log_len, cur_log_batch, unprocessed, cur_code, nxt) == 0
            n_committed += 1
            coset = 1 << (arities[n_committed] if n_committed < len(arities) else p.n_final_challenges())
            rc, nd = o.merkle_build(nxt, coset)
            assert rc == 0
            items.append(("fri_commitment", bytes(nd[-1])))
            cur_code, cur_log_len, cur_log_batch, unprocessed = nxt, new_log_len, 0, []
            next_commit = next_commit + arities[n_committed] if n_committed < len(arities) else None
    items += tail
    terminate = o.arr_to_ints(cur_code)
    items.append(("fri_terminate", terminate))
    return commitment, items, evals, terminate
