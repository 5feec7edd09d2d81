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


@pytest.mark.parametrize("where", ["b_hi", "b_lo", "table", "table_copy", "a_lo"])
@pytest.mark.parametrize("n_vars", [11, 18])
def test_a_copy_into_the_shadows_arrays_ends_the_shadow(oracle, monkeypatch, where, n_vars):
    """between an evaluation and the caller's fold, a copy_d2d INTO an array the shadow describes -- a
    half of b, a half of a, the indicator table, or the buffer the table's lower half has just been copied to -- must not leave
    S = lambda * b (.) eq stale: everything deferred runs in issue order, the side stream is joined, the shadow ends and the
    literal kernels answer.  Two rounds of the literal call sequence with the foreign copy in the first, every returned value
    and, at the end, the caller's arrays against a host model driven by the oracle."""
    import binius_amd
    from binius_amd.sumcheck import bivariate_product_eq_expr, calculate_round_evals

    n = 1 << n_vars
    a, b = oracle.random_b128(0x5AD70000 + n_vars, n), oracle.random_b128(0x5AD70100 + n_vars, n)
    eq = oracle.random_b128(0x5AD70200 + n_vars, n // 2)
    junk = oracle.random_b128(0x5AD70300 + n_vars, 64)
    zs = oracle.random_scalars(0x5AD7, 4)
    with binius_amd.Context(0, 6 * n) as hal:
        alloc = hal.dev_alloc()
        da, db, deq, dj = (upload(hal, alloc, x) for x in (a, b, eq, junk))
        dcopy = alloc.alloc(n // 4)
        e3 = bivariate_product_eq_expr(hal, 0, 1, 2)
        cur, eq_len = n, n // 2
        c0 = hal.arm_counters()
        for r in range(3):
            got = calculate_round_evals(hal, int(np.log2(cur)), [1], [da.slice(0, cur), db.slice(0, cur)], [e3], eq_ind=deq.slice(0, eq_len))
            rc, want = oracle.round_evals_eq([a[:cur].copy(), b[:cur].copy()], int(np.log2(cur)), eq[:eq_len].copy(), [(0, 1)], 1)
            assert rc == 0 and got == want, (where, r)
            half = cur // 2
            if r == 0:
                # the foreign write, between execute and fold
                if where == "b_hi":
                    hal.copy_d2d(dj, db.slice(half + 8, half + 72))
                    b[half + 8 : half + 72] = junk
                elif where == "b_lo":
                    hal.copy_d2d(dj, db.slice(3, 67))
                    b[3:67] = junk
                elif where == "a_lo":
                    hal.copy_d2d(dj, da.slice(5, 69))
                    a[5:69] = junk
                elif where == "table":
                    hal.copy_d2d(dj, deq.slice(1, 65))
                    eq[1:65] = junk
            hal.extrapolate_line_batch([da.slice(0, half), db.slice(0, half)], [da.slice(half, cur), db.slice(half, cur)], zs[r])
            for x in (a, b):
                f = x[:half].copy()
                assert oracle.extrapolate_line(f, x[half:cur].copy(), zs[r]) == 0
                x[:half] = f
            h = eq_len // 2
            if r == 0 and where == "table_copy":
                # the prover's "copy the lower half, add the upper half onto the copy" -- with a foreign write into the copy in between
                hal.copy_d2d(deq.slice(0, h), dcopy.slice(0, h))
                hal.copy_d2d(dj, dcopy.slice(2, 66))
                model_copy = eq[:h].copy()
                model_copy[2:66] = junk

                def k(ke, log_chunks, bufs, h=h):
                    ke.add_assign(int(np.log2(h)) - log_chunks, bufs[1].to_ref(), bufs[0])

                hal.map_kernels(k, [("chunked_mut", dcopy.slice(0, h), 0), ("chunked", deq.slice(h, eq_len), 0)])
                model_copy ^= eq[h:eq_len]
                # the caller goes on with the copy as its table
                eq = model_copy
                deq = dcopy
            else:
                def k(ke, log_chunks, bufs, h=h):
                    ke.add_assign(int(np.log2(h)) - log_chunks, bufs[1].to_ref(), bufs[0])

                hal.map_kernels(k, [("chunked_mut", deq.slice(0, h), 0), ("chunked", deq.slice(h, eq_len), 0)])
                eq[:h] ^= eq[h:eq_len]
            cur, eq_len = half, h
        assert np.array_equal(hal.copy_d2h(da.slice(0, n if da.len >= n else da.len)), a[: da.len])
        assert np.array_equal(hal.copy_d2h(db), b)
        assert np.array_equal(hal.copy_d2h(deq.slice(0, eq_len)), eq[:eq_len])
        c1 = hal.arm_counters()
        if n_vars >= 11:
            assert c1["shadow_dropped"] - c0["shadow_dropped"] >= 1, "the foreign write did not end the shadow"
