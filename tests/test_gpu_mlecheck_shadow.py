"""The literal MLE-check call sequence behind the trait, answered from the bivariate kernels.

An unchanged `BivariateMLEcheckProver` (crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:145-254, 391-520) asks
for sum a * b * eq every round, folds a and b, and folds its indicator table by adding the halves.  The backend keeps a
weighted shadow S = lambda * b (.) eq beside the caller's arrays (csrc/abi_kernels.cpp "shadow") and runs the bivariate
kernels on (a, S); the caller's calls, their results and its buffers are those of the literal kernels -- which still answer
whenever the shadow does not apply (BN_MLECHECK_SHADOW=0, a table that is not a tensor expansion, foreign calls in between).
Everything here goes through the literal prover mirror (BN_MLECHECK=eager) and is compared with the oracle's restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def upload(hal, alloc, arr):
    d = alloc.alloc(arr.shape[0])
    hal.copy_h2d(arr, d)
    return d


def xor_sum(p):
    return int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64)


def _instance(oracle, n_vars, m, comps, seed, table=None):
    mls = [oracle.random_b128(seed + j, 1 << n_vars) for j in range(m)]
    eq_ch = oracle.random_scalars(seed ^ 0xE9, n_vars)
    half = oracle.arr(1 << (n_vars - 1))
    half[0] = (1, 0)
    oracle.tensor_expand(half, 0, eq_ch[: n_vars - 1])
    if table is not None:
        half = table(half)
    sums = []
    for i, j in comps:
        # the claim of the literal prover: sum over the cube of a * b * eq_full, eq_full = tensor of all n coordinates; with a
        # foreign table the transcript is simply whatever the literal operations give -- the oracle runs the same operations
        full = oracle.arr(1 << n_vars)
        full[0] = (1, 0)
        oracle.tensor_expand(full, 0, eq_ch)
        sums.append(xor_sum(oracle.mul_vec(oracle.mul_vec(mls[i], mls[j]), full)))
    stream = oracle.random_scalars(seed ^ 0xC4A2, n_vars + 1)
    return mls, eq_ch, half, sums, stream[0], stream[1:]


def _run(oracle, monkeypatch, n_vars, m, comps, seed, shadow="1", table=None, reps=2):
    import binius_amd
    from binius_amd._host import MlecheckPlan

    monkeypatch.setenv("BN_MLECHECK", "eager")  # the literal trait-call sequence (host/sumcheck.hpp BivariateMLEcheckProver)
    monkeypatch.setenv("BN_MLECHECK_SHADOW", shadow)
    mls, eq_ch, half, sums, bc, ch = _instance(oracle, n_vars, m, comps, seed, table)
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, half.copy(), eq_ch, comps, sums, bc, ch)
    with binius_amd.Context(0, (3 * m + 4) << n_vars) as hal:
        alloc = hal.dev_alloc()
        d = [upload(hal, alloc, x) for x in mls]
        eq_dev = upload(hal, alloc, half)
        scratch = alloc.alloc((m + 1) << (n_vars - 1))
        plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, scratch, comps, sums, bc, ch)
        for _ in range(reps):
            plan.run()
            assert plan.last_mode() == 0  # the literal prover, not the weighted one above the trait
            assert plan.round_coeffs() == want_coeffs
            assert plan.final_evals() == want_finals
        for j in range(m):
            assert np.array_equal(hal.copy_d2h(d[j]), mls[j])  # PreFold inputs are never modified
        assert np.array_equal(hal.copy_d2h(eq_dev), half)
        return hal.arm_counters()


@pytest.mark.parametrize("n_vars", [10, 11, 13, 16, 18, 20])
def test_literal_mlecheck_runs_on_the_shadow(oracle, monkeypatch, n_vars):
    c = _run(oracle, monkeypatch, n_vars, 2, [(0, 1)], 0x5AD00000 + 64 * n_vars)
    # one shadow per prove (two proves), every round down to tables of 2^9 entries... and below: the shadow, once made, serves
    # every later round of the instance
    assert c["shadow_created"] == 2 and c["shadow_dropped"] == 0, c
    assert c["shadow_rounds"] == 2 * n_vars, c


@pytest.mark.parametrize("n_vars", [4, 9, 12])
def test_literal_mlecheck_without_the_shadow(oracle, monkeypatch, n_vars):
    c = _run(oracle, monkeypatch, n_vars, 2, [(0, 1)], 0x5AD10000 + 64 * n_vars, shadow="0")
    assert c["shadow_created"] == 0 and c["shadow_rounds"] == 0


def test_small_instances_keep_the_three_factor_kernel(oracle, monkeypatch):
    c = _run(oracle, monkeypatch, 8, 2, [(0, 1)], 0x5AD20000)
    assert c["shadow_created"] == 0


def test_a_table_that_is_no_tensor_expansion_gets_no_shadow(oracle, monkeypatch):
    """The shadow needs eq[i] == eq[i - 2^k] * rho_k everywhere; one altered entry deep in the table and the literal kernels
    answer (same transcript as the oracle, which multiplies by whatever table it is given)."""
    def spoil(t):
        t = t.copy()
        t[len(t) // 2 + 12345 % (len(t) // 2)] ^= np.uint64(1)
        return t

    c = _run(oracle, monkeypatch, 14, 2, [(0, 1)], 0x5AD30000, table=spoil, reps=1)
    # round 0 is exact whatever the table is (S = b (.) table, element by element); the structure is looked at when the
    # caller's first fold arrives, and there the shadow ends
    assert c["shadow_created"] == 1 and c["shadow_rounds"] == 1 and c["shadow_dropped"] == 1, c

    def scaled(t):  # a constant multiple of a tensor expansion HAS the structure: the shadow applies and stays exact
        k = oracle.random_scalars(0x77, 1)[0]
        kk = oracle.arr(len(t))
        kk[:] = (k & ((1 << 64) - 1), k >> 64)
        return oracle.mul_vec(t, kk)

    c = _run(oracle, monkeypatch, 14, 2, [(0, 1)], 0x5AD30100, table=scaled, reps=1)
    assert c["shadow_created"] == 1 and c["shadow_rounds"] == 14


def test_several_compositions_and_shared_multilinears(oracle, monkeypatch):
    """Only the single-pair launch is shadowed; batched compositions keep the literal kernels -- same transcripts either way."""
    _run(oracle, monkeypatch, 12, 3, [(0, 1), (1, 2)], 0x5AD40000)
    _run(oracle, monkeypatch, 11, 2, [(1, 1)], 0x5AD40100)


def test_round_by_round_with_foreign_calls(oracle, monkeypatch):
    """The handle-based prover (execute / fold one call each) with reads of the caller's arrays between the calls: every read
    flushes what is deferred and ends the shadow, the next evaluation makes a new one, the transcript never changes."""
    import binius_amd
    from binius_amd._host import MlecheckProver

    monkeypatch.setenv("BN_MLECHECK", "eager")
    n_vars, m, comps = 13, 2, [(0, 1)]
    mls, eq_ch, half, sums, bc, ch = _instance(oracle, n_vars, m, comps, 0x5AD50000)
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, half.copy(), eq_ch, comps, sums, bc, ch)
    with binius_amd.Context(0, 16 << n_vars) as hal:
        alloc = hal.dev_alloc()
        d = [upload(hal, alloc, x) for x in mls]
        eq_dev = upload(hal, alloc, half)
        scratch = alloc.alloc((m + 1) << (n_vars - 1))
        prover = MlecheckProver(hal, n_vars, d, eq_dev, eq_ch, scratch, comps, sums)
        got = []
        for r in range(n_vars):
            got.append(prover.execute(bc))
            if r in (1, 4):
                hal.copy_d2h(d[0].slice(0, 1))
            prover.fold(ch[r])
            if r in (2, 4, 7):
                hal.copy_d2h(scratch.slice(0, 1))
        assert got == want_coeffs
        assert prover.finish() == want_finals
        c = hal.arm_counters()
        assert c["shadow_dropped"] >= 3 and c["shadow_created"] >= 2
        prover.close()
