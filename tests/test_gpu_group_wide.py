"""GPU parity of claim groups at the width of the reference's real PCS provers: piop::prove puts ALL committed
multilinears of one size and their transparents into ONE BivariateSumcheckProver (crates/core/src/piop/prove.rs:262-287); the
keccak gadget alone commits 25 state_in + 25 state_out columns of one size plus intermediates
(crates/m3/src/gadgets/hash/keccak/stacked.rs:105,292): k >= 50 claims over m >= 100 multilinears.  The group path
(csrc/abi_group.cpp, kernels_group.hip) carries up to 256 multilinears and 384 claims per prover -- job table, accumulator slots
and value mailbox in memory instead of the kernel-argument block.  Every transcript against the oracle's, bit for bit; the fast
path is asserted from the context's counters (one group launch per round, no plain fold launches)."""
import numpy as np
import pytest

from test_gpu_group import _threads, claim_sums, claims_for, env, oracle_single, upload

pytestmark = pytest.mark.gpu

_ORACLE = {}

WIDE = [(33, "disjoint"), (50, "disjoint"), (64, "disjoint"), (50, "piop")]


def shape(kind, k):
    """(m, comps).  'star': k committed multilinears against ONE transparent.  'keccak': piop::prove's prover for the keccak table --
    k committed columns of one size (100: 25 state_in + 3 x 25 state_out, m3/src/gadgets/hash/keccak/stacked.rs:105,292) against
    three transparents (one per evaluation point the evalcheck reduction leaves: the zerocheck point and those of the shifted
    columns' sumchecks), every column at the first point, half of them at a second, a quarter at a third."""
    if kind == "star":
        return k + 1, [(i, k) for i in range(k)]
    if kind == "keccak":
        c, t = k, 3
        comps = [(i, c + i % t) for i in range(c)] + [(i, c + (i + 1) % t) for i in range(0, c, 2)] + [(i, c + (i + 2) % t) for i in range(0, c, 4)]
        return c + t, comps
    return claims_for(kind, k)


def _run_single(oracle, n_vars, m, comps, seed, group, chain_all=False, ht=None):
    import binius_amd
    from binius_amd._host import SumcheckPlan

    n = 1 << n_vars
    mls = [oracle.random_b128(seed + j, n) for j in range(m)]
    stream = oracle.random_scalars(seed ^ 0x5A5A, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    key = ("wide", n_vars, m, tuple(comps), seed)
    if key not in _ORACLE:
        sums = claim_sums(oracle, mls, comps)
        _ORACLE[key] = (sums, oracle_single(oracle, mls, n_vars, comps, sums, batch_coeff, challenges))
    sums, (want_coeffs, want_final) = _ORACLE[key]
    with env(BN_GROUP=group, BN_GROUP_CHAIN_MIN_LOG2=0 if chain_all else None, BN_GROUP_HT_MAX_LOG2=ht):
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
            for j in (0, m // 2, m - 1):
                assert np.array_equal(hal.copy_d2h(d[j].slice(0, min(n, 4096))), mls[j][: min(n, 4096)]), "an input multilinear was modified"
    for r in range(n_vars):
        assert list(got[0][r]) == list(want_coeffs[r]), "round %d differs from the oracle" % r
    assert list(got[1]) == list(want_final)
    assert again == got
    return cnt


@pytest.mark.parametrize("chain_all", [False, True])
@pytest.mark.parametrize("k,kind,n_vars", [(40, "star", 12), (40, "star", 18), (100, "keccak", 12), (100, "keccak", 16), (100, "keccak", 18), (36, "keccak", 21)])
def test_shared_transparents_vs_oracle(oracle, k, kind, n_vars, chain_all):
    """The claim graph piop::prove builds: every committed multilinear against the few transparents of its size.  One kind-0 job per
    transparent folds it with one column, every other column folds itself on the way and reads the transparent as it is (kind 4),
    further claims of a column are evaluations (kind 1) -- 175 claims over 103 multilinears in ONE launch per round (packed: more
    jobs than the chip has compute units)."""
    m, comps = shape(kind, k)
    cnt = _run_single(oracle, n_vars, m, comps, 0x5A4E0000 + 131 * n_vars + k, group=1, chain_all=chain_all)
    assert cnt["evals"] == n_vars and cnt["launches"] + cnt["hosted_evals"] == n_vars, cnt
    if chain_all:
        assert cnt["prefolds"] == 0 and cnt["jobs_fused"] >= (cnt["launches"] - 1) * (m - 3 if kind == "keccak" else k), cnt


@pytest.mark.parametrize("n_vars", [12, 18, 20])
@pytest.mark.parametrize("k,kind", WIDE)
def test_wide_prover_vs_oracle(oracle, n_vars, k, kind):
    """One BivariateSumcheckProver with 33 / 50 / 64 claims over 66 / 100 / 128 multilinears (disjoint, and the PIOP shape: a shared
    multilinear and one in no claim): all round polynomials and final evaluations equal the oracle's, and every execute() was
    answered on the group path -- ONE launch per round while the arrays are large, host arithmetic once they are small."""
    m, comps = claims_for(kind, k)
    cnt = _run_single(oracle, n_vars, m, comps, 0x71DE0000 + 131 * n_vars + k, group=1)
    assert cnt["evals"] == n_vars and cnt["launches"] + cnt["hosted_evals"] == n_vars, cnt
    if kind == "disjoint":
        assert cnt["jobs_fused"] == k * max(0, cnt["launches"] - 1) and cnt["chains"] == 0 and cnt["prefolds"] == 0, cnt
    assert cnt["flushed_folds"] <= 1, cnt


@pytest.mark.parametrize("k,kind,n_vars", [(50, "piop", 12), (64, "disjoint", 12), (33, "disjoint", 16)])
def test_wide_prover_chained_everywhere(oracle, k, kind, n_vars):
    """The same with the jobs that depend on each other chained inside the launch at every size, and with hosted sessions off
    (every round a launch, down to one evaluation point per claim)."""
    m, comps = claims_for(kind, k)
    cnt = _run_single(oracle, n_vars, m, comps, 0x71DF0000 + 131 * n_vars + k, group=1, chain_all=True, ht=0)
    assert cnt["evals"] == n_vars and cnt["launches"] == n_vars and cnt["prefolds"] == 0 and cnt["hosted_evals"] == 0, cnt


@pytest.mark.parametrize("k,kind", [(50, "piop"), (64, "disjoint")])
def test_wide_prover_eager_path(oracle, k, kind):
    """BN_GROUP=0: the same provers on the single-claim machinery / the eager kernels (what a too-wide prover falls back to)."""
    m, comps = claims_for(kind, k)
    cnt = _run_single(oracle, 12, m, comps, 0x71DE0000 + 131 * 12 + k, group=0)
    assert cnt["launches"] == 0, cnt


@pytest.mark.parametrize("seed", list(range(8)))
def test_wide_claim_graph_fuzz(oracle, seed):
    """Random claim graphs of up to 60 claims over up to 80 multilinears -- claims sharing multilinears, multilinears in no claim, the
    occasional square, stars (one multilinear against many) -- at 2^10 .. 2^17 elements per array; odd seeds chain dependent jobs
    inside the launch at every size."""
    rng = np.random.RandomState(0x7A1D + 104729 * seed)
    m = int(rng.randint(34, 81))
    k = int(rng.randint(20, 61))
    n_vars = int(rng.randint(10, 18))
    comps = []
    hub = int(rng.randint(m))
    for c in range(k):
        r = rng.rand()
        if r < 0.15:
            i, j = hub, int(rng.randint(m))  # a star around one multilinear
        elif r < 0.2:
            i = j = int(rng.randint(m))  # a square
        else:
            i, j = int(rng.randint(m)), int(rng.randint(m))
        comps.append((i, j))
    comps = list(dict.fromkeys(comps))  # (the prover's claims are distinct compositions)
    cnt = _run_single(oracle, n_vars, m, comps, 0x7A1D0000 + 4096 * seed, group=1, chain_all=bool(seed % 2))
    assert cnt["evals"] == n_vars and cnt["launches"] + cnt["hosted_evals"] == n_vars, cnt


@pytest.mark.parametrize("group", [1, 0])
def test_too_wide_prover_beside_pending_group_folds(oracle, group):
    """A prover BEYOND the group path's width (260 multilinears; 130 claims) in a front-loaded batch with two ordinary provers:
    its evaluations take the eager path while the others' folds are deferred on the group path -- and its own fold arrives in two
    calls (256 + 4 slices).  Round proofs and final evaluations of the whole batch equal the oracle's; a second prove repeats them."""
    import binius_amd
    from binius_amd._host import BatchSumcheckPlan
    from oracle import piop_ref

    shapes = [(9, 3, "piop"), (10, 130, "disjoint"), (12, 2, "disjoint")]
    provers = []
    for p, (v, k, kind) in enumerate(shapes):
        m, comps = claims_for(kind, k)
        mls = [oracle.random_b128(0x700F0000 + 0x1000 * p + j, 1 << v) for j in range(m)]
        provers.append((v, mls, comps, claim_sums(oracle, mls, comps)))
    sizes = [v for v, _, _ in shapes]
    stream = oracle.random_scalars(0x700F, len(sizes) + max(sizes))
    batch_coeffs, challenges = stream[: len(sizes)], stream[len(sizes) :]
    total = sum(len(mls) << v for v, mls, _, _ in provers)
    with env(BN_GROUP=group):
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
    ref = [dict(n_vars=v, multilins=[x.copy() for x in mls], comps=comps, sums=sums) for v, mls, comps, sums in provers]
    items, evals = piop_ref.batch_sumcheck_prove(ref, batch_coeffs, challenges, threads=_threads())
    want_proofs = [list(p) + [0] * (2 - len(p)) for k, p in items if k == "round_proof"]
    assert got[0] == want_proofs
    assert got[1] == evals
    if group:
        assert cnt["evals"] > 0, cnt  # (the ordinary provers did run on the group path)


def test_wide_prover_in_a_front_loaded_batch(oracle):
    """A keccak-width prover (50 claims, 100 multilinears) front-loaded with two small ones: from the third batch round on a round is
    ONE launch carrying all three provers' jobs (the wide one's 50 and the riders'), the riders' execute() calls answered from the
    sums computed ahead."""
    import binius_amd
    from binius_amd._host import BatchSumcheckPlan
    from oracle import piop_ref

    shapes = [(15, 2, "piop"), (16, 1, "disjoint"), (18, 50, "piop")]
    provers = []
    for p, (v, k, kind) in enumerate(shapes):
        m, comps = claims_for(kind, k)
        mls = [oracle.random_b128(0x701F0000 + 0x1000 * p + j, 1 << v) for j in range(m)]
        provers.append((v, mls, comps, claim_sums(oracle, mls, comps)))
    sizes = [v for v, _, _ in shapes]
    stream = oracle.random_scalars(0x701F, len(sizes) + max(sizes))
    batch_coeffs, challenges = stream[: len(sizes)], stream[len(sizes) :]
    total = sum(len(mls) << v for v, mls, _, _ in provers)
    with binius_amd.Context(0, total + total // 2 + 4096) as hal:
        alloc = hal.dev_alloc()
        dev = [(v, [upload(hal, alloc, x) for x in mls], comps, sums) for v, mls, comps, sums in provers]
        scratch = alloc.alloc(total // 2 + 64)
        plan = BatchSumcheckPlan(hal, dev, scratch, batch_coeffs, challenges)
        plan.run()
        got = (plan.round_proofs(), plan.final_evals())
        cnt = hal.group_counters()
    ref = [dict(n_vars=v, multilins=[x.copy() for x in mls], comps=comps, sums=sums) for v, mls, comps, sums in provers]
    items, evals = piop_ref.batch_sumcheck_prove(ref, batch_coeffs, challenges, threads=_threads(), fast=True)
    want_proofs = [list(p) + [0] * (2 - len(p)) for k, p in items if k == "round_proof"]
    assert got[0] == want_proofs
    assert got[1] == evals
    # (the provers start together, front_loaded.rs:122-155; a small one is answered on the host from 2^12 elements per array on, the
    # wide one -- 100 arrays -- from 2^10: while two of them are on the device, the second's sums ride in the first's launch)
    assert cnt["spec_hits"] > 0 and cnt["spec_jobs"] >= 50 and cnt["launches"] <= max(sizes) + 3 * len(sizes), cnt
