"""GPU parity of the call shape the reference's PCS prover issues: k product claims over m multilinears in
ONE BivariateSumcheckProver -- multilinears shared between claims, a multilinear in no claim, a batch coefficient != 1 --,
several provers front-loaded on one context (protocols/sumcheck/prove/front_loaded.rs:122-155), and piop::prove with FRI
interleaved (crates/core/src/piop/prove.rs:148-395), at the sizes where the dispatcher runs the matrix-core kernels -- every
transcript against the oracle's, bit for bit, with the claim-group path on (csrc/abi_group.cpp, kernels_group.hip) and off
(BN_GROUP=0: the single-claim machinery and the eager kernels)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


class env:
    """environment switches that bn_ctx_create reads, for the contexts created inside the block"""

    def __init__(self, **kv):
        self.kv = {k: v for k, v in kv.items() if v is not None}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def upload(hal, alloc, arr):
    d = alloc.alloc(arr.shape[0])
    chunk = 1 << 22
    for off in range(0, arr.shape[0], chunk):
        hal.copy_h2d(arr[off : off + chunk], d.slice(off, min(arr.shape[0], off + chunk)))
    return d


def claims_for(kind, k):
    """(m, comps) of a k-claim prover.  'piop': committed 0 .. k-1, transparents k .. 2k-1, claim i = (committed i, transparent i)
    except that the last claim re-uses committed 0 and committed k-1 is in no claim (an unconstrained column) -- a shared array,
    an array in no claim, and k - 2 disjoint claims (piop/verify.rs:240-268).  'disjoint': (i, k + i).  'bipartite': every
    committed against every transparent (compute_test_utils/src/piop.rs:60-99), k = c * c."""
    if kind == "disjoint":
        return 2 * k, [(i, k + i) for i in range(k)]
    if kind == "piop":
        comps = [(i, k + i) for i in range(k - 1)] + [(0, 2 * k - 1)]
        return 2 * k, comps
    c = int(round(k ** 0.5))
    assert c * c == k
    return 2 * c, [(i, c + j) for i in range(c) for j in range(c)]


def oracle_single(oracle, mls, n_vars, comps, sums, batch_coeff, challenges):
    ref = [x.copy() for x in mls]
    if n_vars >= 16:
        got = oracle.fast_bivariate_sumcheck_prove(ref, n_vars, comps, sums, batch_coeff, challenges, threads=_threads())
        if got is not None:
            return got
        ref = [x.copy() for x in mls]
    return oracle.bivariate_sumcheck_prove(ref, n_vars, comps, sums, batch_coeff, challenges, threads=_threads())


def claim_sums(oracle, mls, comps):
    out = []
    for i, j in comps:
        s = oracle.fast_inner_product(mls[i], mls[j], _threads()) if mls[i].shape[0] >= (1 << 16) else None
        if s is None:
            rc, s = oracle.inner_product(mls[i], 7, mls[j])
            assert rc == 0
        out.append(s)
    return out


_ORACLE = {}
CASES = [(12, 2, "piop"), (12, 4, "piop"), (12, 8, "piop"), (12, 4, "bipartite"), (18, 2, "piop"), (18, 4, "disjoint"), (18, 8, "piop"), (20, 2, "disjoint"),
         (20, 4, "piop"), (20, 4, "bipartite"), (20, 8, "disjoint"), (22, 2, "piop"), (22, 4, "disjoint"), (9, 3, "piop"), (5, 2, "disjoint"), (3, 2, "piop"),
         (23, 4, "bipartite"),  # (large enough for the default to chain the first rounds' jobs)
         (24, 4, "disjoint")]  # (the shape bench.py times beside the headline as `claim_groups`, at its size)


@pytest.mark.parametrize("group", [2, 1, 0])
@pytest.mark.parametrize("n_vars,k,kind", CASES)
def test_multi_claim_prover_vs_oracle(oracle, n_vars, k, kind, group):
    """One BivariateSumcheckProver with k claims (SumcheckPlan = the C++ mirror's execute / fold / finish loop): all round
    polynomials and final evaluations equal the oracle's; the inputs are untouched; with the group path on, every round after
    the first is ONE launch that folds and evaluates.  group = 2: jobs that depend on each other are chained inside the launch
    at every size (the library's default does that from 2^21 evaluation points per claim, where it pays); group = 1: the
    default -- at these sizes the shared arrays are folded by a plain launch in front."""
    import binius_amd
    from binius_amd._host import SumcheckPlan

    if group == 0 and n_vars > 20:
        pytest.skip("the eager path at this size is covered by (20, *)")
    m, comps = claims_for(kind, k)
    n = 1 << n_vars
    mls = [oracle.random_b128(0x6A0B0000 + 97 * n_vars + j, n) for j in range(m)]
    stream = oracle.random_scalars(0x6A0C + n_vars + k, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    assert batch_coeff not in (0, 1)
    key = ("claims", n_vars, k, kind)
    if key not in _ORACLE:  # (the oracle's side of a case is computed once for the group-on and group-off runs)
        sums = claim_sums(oracle, mls, comps)
        _ORACLE[key] = (sums, oracle_single(oracle, mls, n_vars, comps, sums, batch_coeff, challenges))
    sums, (want_coeffs, want_final) = _ORACLE[key]
    with env(BN_GROUP=min(group, 1), BN_GROUP_CHAIN_MIN_LOG2=0 if group == 2 else None):
        with binius_amd.Context(0, m * n + m * (n // 2) + 4096) as hal:
            alloc = hal.dev_alloc()
            d = [upload(hal, alloc, x) for x in mls]
            scratch = alloc.alloc(max(1, m * (n // 2)))
            plan = SumcheckPlan(hal, n_vars, d, scratch, comps, sums, batch_coeff, challenges)
            plan.run()
            got = (plan.round_coeffs(), plan.final_evals())
            cnt = hal.group_counters()
            plan.run()  # a second prove from the same resident inputs
            again = (plan.round_coeffs(), plan.final_evals())
            for j in (0, m - 1):
                assert np.array_equal(hal.copy_d2h(d[j].slice(0, min(n, 4096))), mls[j][: min(n, 4096)])
    for r in range(n_vars):
        assert list(got[0][r]) == list(want_coeffs[r]), "round %d differs from the oracle" % r
    assert list(got[1]) == list(want_final)
    assert again == got
    if group:
        # every execute() was answered on the group path: by a launch of the group kernel while the arrays are large, on the host
        # once they are small (hosted sessions: at most 2^12 elements per array, 2^8 on a host without VPCLMULQDQ)
        assert cnt["evals"] == n_vars and cnt["launches"] + cnt["hosted_evals"] == n_vars, cnt
        if kind == "disjoint":
            assert cnt["jobs_fused"] == k * max(0, cnt["launches"] - 1) and cnt["chains"] == 0 and cnt["prefolds"] == 0, cnt
        elif group == 2:
            # every fold is a job of a group launch, none a launch of its own: claims over arrays of their own fused with the folds,
            # shared arrays folded by the workgroups that then evaluate the claims over them
            assert cnt["prefolds"] == 0, cnt
            if cnt["launches"] > 1:
                assert cnt["chains"] >= cnt["launches"] - 1 and cnt["jobs_fused"] > 0, cnt
        elif n_vars >= 23:
            assert cnt["chains"] >= n_vars - 22 and cnt["prefolds"] > 0, cnt
        else:
            assert cnt["chains"] == 0 and cnt["jobs_fold"] == 0, cnt
        assert cnt["flushed_folds"] <= 1, cnt  # (at most the last fold, forced out by finish()'s reads)
    else:
        assert cnt["launches"] == 0


def batch_instance(oracle, sizes, ks, seed):
    """provers ascending by n_vars: (n_vars, multilins, comps, sums)"""
    provers = []
    for p, (v, k) in enumerate(zip(sizes, ks)):
        m, comps = claims_for("piop" if k >= 2 else "disjoint", k) if k else (2, [])
        mls = [oracle.random_b128(seed + 0x100 * p + j, 1 << v) for j in range(m)]
        provers.append((v, mls, comps, claim_sums(oracle, mls, comps)))
    return provers


@pytest.mark.parametrize("group,spec", [(2, 1), (1, 1), (1, 0), (0, 0)])  # (group = 2: chains at every size, as in the test above)
@pytest.mark.parametrize("sizes,ks", [([9, 9, 11, 12], [2, 1, 1, 3]), ([17, 17, 19, 20], [2, 1, 1, 2]), ([6, 8], [1, 1]), ([4, 13, 13], [0, 2, 1]), ([16, 18], [4, 2])])
def test_front_loaded_batch_vs_oracle(oracle, sizes, ks, group, spec):
    """SumcheckBatchProver::run over several BivariateSumcheckProvers on ONE context, in the reference's order (execute on every
    prover, one challenge, fold on every prover): round proofs and final evaluations against the oracle's restatement
    (oracle/piop_ref.py).  With the group path on, a batch round is ONE launch: the first execute() of the round carries the
    other provers' claims, whose execute() calls are then answered without a launch."""
    import binius_amd
    from binius_amd._host import BatchSumcheckPlan
    from oracle import piop_ref

    provers = batch_instance(oracle, sizes, ks, 0x5EED0000 + sum(sizes))
    stream = oracle.random_scalars(0xBA7C + sum(sizes), len(sizes) + max(sizes))
    batch_coeffs, challenges = stream[: len(sizes)], stream[len(sizes) :]
    total = sum(len(mls) << v for v, mls, _, _ in provers)
    with env(BN_GROUP=min(group, 1), BN_GROUP_SPEC=spec, BN_GROUP_CHAIN_MIN_LOG2=0 if group == 2 else None):
        with binius_amd.Context(0, total + total // 2 + 4096) as hal:
            alloc = hal.dev_alloc()
            dev = [(v, [upload(hal, alloc, x) for x in mls], comps, sums) for v, mls, comps, sums in provers]
            scratch = alloc.alloc(total // 2 + 64)
            plan = BatchSumcheckPlan(hal, dev, scratch, batch_coeffs, challenges)
            plan.run()
            got = (plan.round_proofs(), plan.final_evals())
            cnt = hal.group_counters()
            plan.run()
            assert (plan.round_proofs(), plan.final_evals()) == got
    key = ("batch", tuple(sizes), tuple(ks))
    if key not in _ORACLE:
        ref = [dict(n_vars=v, multilins=[x.copy() for x in mls], comps=comps, sums=sums) for v, mls, comps, sums in provers]
        _ORACLE[key] = piop_ref.batch_sumcheck_prove(ref, batch_coeffs, challenges, threads=_threads(), fast=max(sizes) >= 16)
    items, evals = _ORACLE[key]
    want_proofs = [list(p) + [0] * (2 - len(p)) for k, p in items if k == "round_proof"]
    assert got[0] == want_proofs
    assert got[1] == evals
    if group and spec:
        # every execute() of a prover with claims was answered on the group path; from the third batch round on (the sessions of all
        # provers are known after their first group evaluation) a round costs one launch
        n_exec = sum(v for v, k in zip(sizes, ks) if k)
        assert cnt["evals"] + 0 >= n_exec - 2 * len(sizes), cnt
        assert cnt["spec_hits"] > 0 or cnt["hosted_evals"] > 0 or len([k for k in ks if k]) < 2, cnt
        assert cnt["launches"] <= max(sizes) + 3 * len(sizes), cnt


@pytest.mark.parametrize("group", [2, 1, 0])  # (2: chains at every size)
@pytest.mark.parametrize("n,log_inv_rate,log_batch,arities", [(12, 1, 3, [4, 4]), (20, 1, 4, [4, 4, 4, 4]), (9, 2, 0, [3, 2]), (10, 1, 2, [])])
def test_piop_prove_vs_oracle(oracle, n, log_inv_rate, log_batch, arities, group):
    """piop::prove (bnh_piop_prove = binius_amd/host/piop.hpp): committed multilinears of n-3, n-3, n-1 and n variables
    (CommitMeta::with_vars, the reference's own PIOP test shapes), two transparents per size, a claim for every (committed,
    transparent) pair of equal size (compute_test_utils/src/piop.rs:60-99), FRI interleaved every round: the commitment and the
    whole transcript -- round proofs, final evaluations, FRI round commitments, terminate codeword -- against the oracle's
    restatement."""
    import binius_amd
    from binius_amd._host import FRIParams, PiopPlan
    from oracle import piop_ref

    n_varss = [n - 3, n - 3, n - 1, n]
    meta = piop_ref.CommitMeta.with_vars(n_varss)
    p = FRIParams(meta.total_vars - log_batch, log_inv_rate, log_batch, arities, n_test_queries=3)
    seed = 0x9109 + 977 * n
    committed = [oracle.random_b128(seed + 16 * i, 1 << v) for i, v in enumerate(n_varss)]
    t_sizes = [v for v in sorted(set(n_varss)) for _ in range(2)]
    transparents = [oracle.random_b128(seed + 0x1000 + 16 * j, 1 << v) for j, v in enumerate(t_sizes)]
    claims = []
    for i, c in enumerate(committed):
        for j, t in enumerate(transparents):
            if c.shape[0] == t.shape[0]:
                s = claim_sums(oracle, [c, t], [(0, 1)])[0]
                claims.append((c.shape[0].bit_length() - 1, i, j, s))
    sizes = sorted(set(n_varss))
    stream = oracle.random_scalars(0x7A0 + n, len(sizes) + meta.total_vars)
    batch_coeffs, challenges = stream[: len(sizes)], stream[len(sizes) :]
    message = piop_ref.merge_multilins(committed, meta.total_vars)
    code_elems = 1 << (meta.total_vars + log_inv_rate)
    ml_elems = sum(x.shape[0] for x in committed) + sum(x.shape[0] for x in transparents)
    with env(BN_GROUP=min(group, 1), BN_GROUP_CHAIN_MIN_LOG2=0 if group == 2 else None):
        with binius_amd.Context(0, message.shape[0] + 4 * code_elems + 2 * ml_elems + (1 << 16)) as hal:
            alloc = hal.dev_alloc()
            d_c = [(v, upload(hal, alloc, x)) for v, x in zip(n_varss, committed)]
            d_t = [(v, upload(hal, alloc, x)) for v, x in zip(t_sizes, transparents)]
            d_msg = upload(hal, alloc, message)
            scratch = alloc.alloc(4 * code_elems + ml_elems + (1 << 14))
            plan = PiopPlan(hal, d_c, d_t, claims, p, d_msg, scratch, batch_coeffs, challenges)
            plan.run()
            got_commitment, got = bytes(plan.commitment), plan.transcript()
            cnt = hal.group_counters()
    key = ("piop", n, log_inv_rate, log_batch, tuple(arities))
    if key not in _ORACLE:
        _ORACLE[key] = piop_ref.piop_prove(committed, transparents, claims, p, batch_coeffs, challenges, threads=_threads(), fast=n >= 16)
    commitment, items, evals, terminate = _ORACLE[key]
    assert got_commitment == commitment
    assert [k for k, _ in got] == [k for k, _ in items]
    for idx, ((k, a), (_, b)) in enumerate(zip(got, items)):
        assert a == b, "transcript item %d (%s) differs from the oracle" % (idx, k)
    if group:
        # fused launches on the rounds between FRI commitments: the commit rounds' foreign calls (fri_fold, the Merkle tree, its root)
        # may force the deferred folds out; no other round does
        assert cnt["launches"] + cnt["hosted_evals"] > 0 and cnt["flushed_folds"] <= 3 * (len(arities) + len(sizes)) + 2, cnt


@pytest.mark.parametrize("ht_log2", [0, 3, 6, 12])
def test_hosted_sessions_any_threshold(oracle, ht_log2):
    """Hosted sessions (csrc/abi_group.cpp): once a prover's arrays are at most 2^ht_log2 elements its remaining rounds are host
    arithmetic on copies handed over by one launch; the device catches up by one write-back launch when its memory is looked at.
    Whatever the threshold (0 = off), the transcript of a three-prover batch is the oracle's, a second prove from the same
    inputs repeats it, the inputs are untouched and the folded buffers end up byte for byte as with the threshold off."""
    import binius_amd
    from binius_amd._host import BatchSumcheckPlan
    from oracle import piop_ref

    sizes, ks = [7, 10, 13], [2, 1, 3]
    provers = batch_instance(oracle, sizes, ks, 0x405F0000)
    stream = oracle.random_scalars(0x405F, len(sizes) + max(sizes))
    batch_coeffs, challenges = stream[: len(sizes)], stream[len(sizes) :]
    total = sum(len(mls) << v for v, mls, _, _ in provers)
    dumps = []
    for ht in (ht_log2, 0):
        with env(BN_GROUP_HT_MAX_LOG2=ht):
            with binius_amd.Context(0, total + total // 2 + 4096) as hal:
                alloc = hal.dev_alloc()
                dev = [(v, [upload(hal, alloc, x) for x in mls], comps, sums) for v, mls, comps, sums in provers]
                scratch = alloc.alloc(total // 2 + 64)
                plan = BatchSumcheckPlan(hal, dev, scratch, batch_coeffs, challenges)
                plan.run()
                got = (plan.round_proofs(), plan.final_evals())
                cnt = hal.group_counters()
                plan.run()
                assert (plan.round_proofs(), plan.final_evals()) == got
                dumps.append((got, hal.copy_d2h(scratch), [hal.copy_d2h(d) for _, ds, _, _ in dev for d in ds], cnt))
    ref = [dict(n_vars=v, multilins=[x.copy() for x in mls], comps=comps, sums=sums) for v, mls, comps, sums in provers]
    items, evals = piop_ref.batch_sumcheck_prove(ref, batch_coeffs, challenges)
    want_proofs = [list(p) + [0] * (2 - len(p)) for k, p in items if k == "round_proof"]
    assert dumps[0][0] == (want_proofs, evals) and dumps[1][0] == dumps[0][0]
    assert np.array_equal(dumps[0][1], dumps[1][1]), "the folded buffers differ from what execution without hosted sessions leaves"
    for a, b, (v, mls, _, _) in zip(dumps[0][2], dumps[1][2], [(v, x, 0, 0) for v, mls, _, _ in provers for x in mls]):
        assert np.array_equal(a, b) and np.array_equal(a, mls)
    if ht_log2 >= 3 and dumps[0][3]["hosted_started"] == 0:
        arm = None
        with binius_amd.Context(0, 4096) as hal:
            arm = hal.arm_counters()
        assert arm["ht_max"] == 0, "hosted sessions never started although the host tail is available"
    assert dumps[1][3]["hosted_started"] == 0


@pytest.mark.parametrize("kind,chain_min", [("bipartite", None), ("piop", None), ("bipartite", 63)])
def test_shared_claims_at_size_satisfy_the_verifier(kind, chain_min):
    """The shapes with shared multilinears at 2^26 elements per array (4 - 8 GiB resident; inputs generated on the device: tensor
    expansions of random points), where the library's default chains the first rounds' jobs inside the launch -- too large for the
    oracle in test time, so the size-independent property is checked on the transcript the device produced: every round polynomial
    satisfies P_r(0) + P_r(1) = the running claim (the claims from the device's own inner products), and the batched product of
    the final evaluations is the last running sum -- the sumcheck verifier's equations (protocols/sumcheck/verify.rs), which tie
    every round evaluation, every fold and the final reads together.  chain_min = 63: the same with plain fold launches."""
    import binius_amd
    from binius_amd import synthetic
    from binius_amd._host import SumcheckPlan

    F = binius_amd.HostField
    n_vars, k = 26, 4
    m, comps = claims_for(kind, k)
    n = 1 << n_vars
    with env(BN_GROUP_CHAIN_MIN_LOG2=chain_min):
        with binius_amd.Context(0, m * n + m * (n // 2) + 4096) as hal:
            alloc = hal.dev_alloc()
            d = []
            for j in range(m):
                x = alloc.alloc(n)
                hal.fill(x.slice(0, 1), 0x5A + j)
                hal.tensor_expand(0, synthetic.random_scalars(0x26C0 + j, n_vars), x)
                d.append(x)
            scratch = alloc.alloc(m * (n // 2))
            sums = [hal.inner_product(d[i], 7, d[j]) for i, j in comps]
            stream = synthetic.random_scalars(0x26C1, n_vars + 1)
            bc, ch = stream[0], stream[1:]
            plan = SumcheckPlan(hal, n_vars, d, scratch, comps, sums, bc, ch)
            plan.run()
            coeffs, finals = plan.round_coeffs(), plan.final_evals()
            cnt = hal.group_counters()

    def horner(cs, x):
        e = 0
        for c in reversed(cs):
            e = F.mul(e, x) ^ c
        return e

    running = horner(sums, bc)
    for r in range(n_vars):
        c = coeffs[r]
        assert (c[0] ^ (c[0] ^ c[1] ^ c[2])) == running, "round %d: P(0) + P(1) is not the running claim" % r
        running = horner(c, ch[r])
    acc, p = 0, 1
    for i, j in comps:
        acc ^= F.mul(p, F.mul(finals[i], finals[j]))
        p = F.mul(p, bc)
    assert acc == running, "the final evaluations do not match the last round polynomial"
    if chain_min is None:
        assert cnt["chains"] >= 4 and cnt["jobs_fused"] > 0, cnt  # (rounds with 2^25 ... 2^22 points per claim... are chained)
    else:
        assert cnt["chains"] == 0 and cnt["prefolds"] > 0, cnt


def test_more_hosted_provers_than_sessions_keep_their_folds(oracle):
    """A context remembers a bounded number of provers (sessions; the least recently used one is forgotten).  A prover that finished
    on the host may still be owed the write-back of its folded arrays -- nobody has looked at that memory yet: forgetting it must
    not lose those folds.  Twenty small provers, one after the other on one context, each finishing on the host; only then is every
    folded buffer read: byte for byte what eager execution leaves, and every transcript the oracle's."""
    import binius_amd
    from binius_amd._host import SumcheckPlan

    n_provers, n_vars = 20, 9
    m, comps = claims_for("piop", 3)
    n = 1 << n_vars
    insts = []
    for p in range(n_provers):
        mls = [oracle.random_b128(0x5E550000 + 64 * p + j, n) for j in range(m)]
        sums = claim_sums(oracle, mls, comps)
        stream = oracle.random_scalars(0x5E55 + p, n_vars + 1)
        insts.append((mls, sums, stream[0], stream[1:]))
    dumps = []
    for lazy in (1, 0):
        with env(BN_NO_LAZY_FOLD=None if lazy else 1):
            with binius_amd.Context(0, n_provers * (m * n + m * (n // 2)) + 4096) as hal:
                alloc = hal.dev_alloc()
                outs, bufs = [], []
                for mls, sums, bc, ch in insts:
                    d = [upload(hal, alloc, x) for x in mls]
                    scratch = alloc.alloc(m * (n // 2))
                    plan = SumcheckPlan(hal, n_vars, d, scratch, comps, sums, bc, ch)
                    plan.run()
                    outs.append((plan.round_coeffs(), plan.final_evals()))
                    bufs.append((d, scratch))
                cnt = hal.group_counters()
                mem = [(hal.copy_d2h(scratch), [hal.copy_d2h(x) for x in d]) for d, scratch in bufs]
                dumps.append((outs, mem, cnt))
    for p, (mls, sums, bc, ch) in enumerate(insts):
        want = oracle_single(oracle, mls, n_vars, comps, sums, bc, ch)
        for which in (0, 1):
            got = dumps[which][0][p]
            assert [list(r) for r in got[0]] == [list(r) for r in want[0]] and list(got[1]) == list(want[1]), ("transcript", which, p)
        assert np.array_equal(dumps[0][1][p][0], dumps[1][1][p][0]), "prover %d: folded buffers differ from what eager execution leaves" % p
        for a, b, x in zip(dumps[0][1][p][1], dumps[1][1][p][1], mls):
            assert np.array_equal(a, b) and np.array_equal(a, x)
    c = dumps[0][2]
    if c["hosted_started"]:  # (a host without carry-less multiplication hosts nothing at this size)
        assert c["hosted_started"] >= n_provers and c["hosted_writebacks"] >= n_provers - 16, c


@pytest.mark.parametrize("group", [1, 0])
@pytest.mark.parametrize("n_varss,n_transparents,log_inv_rate", __import__("piop_verify").REFERENCE_SUITE)
def test_reference_piop_suite(oracle, n_varss, n_transparents, log_inv_rate, group):
    """Port of the reference's PIOP tests (crates/core/tests/piop.rs:10-112 -> compute_test_utils/src/piop.rs:101
    commit_prove_verify): commit, prove through the backend (bnh_piop_prove: the C++ mirror of piop::prove over the C ABI), then
    VERIFY -- piop::verify's equations on the device's own transcript (tests/piop_verify.py) -- for one polynomial, no opening
    claims at all (provers without a composition), one size, an extreme rate (log_inv_rate 8), the small and the standard mix;
    FRI parameters chosen as make_commit_params_with_optimal_arity chooses them.  And the transcript equals the oracle's."""
    import binius_amd
    from binius_amd._host import FRIParams, PiopPlan
    from oracle import piop_ref
    from piop_verify import make_instance, optimal_params, verify_transcript

    meta = piop_ref.CommitMeta.with_vars(n_varss)
    op = optimal_params(meta.total_vars, log_inv_rate)
    p = FRIParams(op.log_dim, op.log_inv_rate, op.log_batch_size, op.fold_arities, n_test_queries=op.n_test_queries)
    committed, transparents, claims = make_instance(oracle, n_varss, n_transparents, 0x51ED + sum(n_varss))
    t_sizes = [t.shape[0].bit_length() - 1 for t in transparents]
    n_sizes = len(set(n_varss))
    stream = oracle.random_scalars(0x7A0 + len(n_varss), n_sizes + meta.total_vars)
    batch_coeffs, challenges = stream[:n_sizes], stream[n_sizes:]
    message = piop_ref.merge_multilins(committed, meta.total_vars)
    code_elems = 1 << (meta.total_vars + log_inv_rate)
    ml_elems = sum(x.shape[0] for x in committed) + sum(x.shape[0] for x in transparents)
    with env(BN_GROUP=min(group, 1), BN_GROUP_CHAIN_MIN_LOG2=0 if group == 2 else None):
        with binius_amd.Context(0, message.shape[0] + 4 * code_elems + 2 * ml_elems + (1 << 16)) as hal:
            alloc = hal.dev_alloc()
            d_c = [(v, upload(hal, alloc, x)) for v, x in zip(n_varss, committed)]
            d_t = [(v, upload(hal, alloc, x)) for v, x in zip(t_sizes, transparents)]
            d_msg = upload(hal, alloc, message)
            scratch = alloc.alloc(4 * code_elems + ml_elems + (1 << 14))
            plan = PiopPlan(hal, d_c, d_t, claims, p, d_msg, scratch, batch_coeffs, challenges)
            plan.run()
            got_commitment, got = bytes(plan.commitment), plan.transcript()
            for (v, d), x in zip(d_c + d_t, committed + transparents):
                assert np.array_equal(hal.copy_d2h(d), x), "an input multilinear was modified"
    verify_transcript(oracle, piop_ref, n_varss, committed, transparents, claims, op, batch_coeffs, challenges, got)
    commitment, items, evals, terminate = piop_ref.piop_prove(committed, transparents, claims, op, batch_coeffs, challenges)
    assert got_commitment == commitment and got == items
