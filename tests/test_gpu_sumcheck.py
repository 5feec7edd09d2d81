"""GPU parity tests for the sumcheck path: round evaluation (accumulate_kernels), fold
(extrapolate_line) and the full prover loop, vs the CPU oracle -- bit-exact.

Mirrors crates/compute_test_utils/src/bivariate_sumcheck.rs:44-262
(generic_test_calculate_round_evals, generic_test_bivariate_sumcheck_prove_verify) with the
transcript replaced by a seeded challenge stream; the "verifier" checks are the sumcheck
verifier's own: P(0) + P(1) == running sum every round, and the final evaluations equal the
multilinear extensions at the reversed challenge point (bivariate_sumcheck.rs:257-261).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, 1 << 23)
    yield ctx
    ctx.close()


def upload(hal, alloc, arr):
    d = alloc.alloc(arr.shape[0])
    hal.copy_h2d(arr, d)
    return d


def powers(oracle, x, n):
    out, p = [], 1
    for _ in range(n):
        out.append(p)
        p = oracle.mul(p, x)
    return out


@pytest.mark.parametrize("n_vars,m,n_comps", [(1, 2, 1), (2, 2, 1), (5, 3, 2), (8, 8, 8), (11, 2, 1), (13, 4, 3)])
def test_calculate_round_evals(hal, oracle, n_vars, m, n_comps):
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(0xB1A50000 + j, 1 << n_vars) for j in range(m)]
    d = [upload(hal, alloc, x) for x in mls]
    rng = np.random.RandomState(n_vars * 100 + m)
    comps = [(int(rng.randint(m)), int(rng.randint(m))) for _ in range(n_comps)]
    batch_coeff = oracle.random_scalars(0xC4A1, 1)[0]
    exprs = [bivariate_product_expr(hal, i, j) for i, j in comps]
    got = calculate_round_evals(hal, n_vars, powers(oracle, batch_coeff, n_comps), d, exprs)
    rc, want = oracle.round_evals(mls, n_vars, comps, batch_coeff)
    assert rc == 0
    assert got == want


@pytest.mark.parametrize("n_vars,m,n_comps", [(1, 2, 1), (3, 2, 1), (8, 8, 8), (12, 2, 1), (16, 2, 1), (14, 3, 2)])
def test_bivariate_sumcheck_prove(hal, oracle, n_vars, m, n_comps):
    """generic_test_bivariate_sumcheck_prove_verify (compute_test_utils bivariate_sumcheck.rs): the prover mirror
    (binius_amd/host/sumcheck.hpp) over the HIP backend, transcript against the oracle and the verifier's checks."""
    from binius_amd._host import SumcheckPlan

    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(0xB1A50000 + j, 1 << n_vars) for j in range(m)]
    d = [upload(hal, alloc, x) for x in mls]
    rng = np.random.RandomState(n_vars * 7 + m)
    comps = [(int(rng.randint(m)), int(rng.randint(m))) for _ in range(n_comps)]
    sums = []
    for i, j in comps:
        rc, s = oracle.inner_product(mls[i], 7, mls[j])
        assert rc == 0
        sums.append(s)
    stream = oracle.random_scalars(0xC4A1, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]

    plan = SumcheckPlan(hal, n_vars, d, alloc.alloc(max(1, m << max(0, n_vars - 1))), comps, sums, batch_coeff, challenges)
    plan.run()
    got_coeffs, final = plan.round_coeffs(), plan.final_evals()
    running = oracle.evaluate_univariate(sums, batch_coeff)
    for r in range(n_vars):
        rc = got_coeffs[r]
        # verifier check: P(0) + P(1) == running sum
        assert rc[0] ^ (rc[0] ^ rc[1] ^ rc[2]) == running
        running = oracle.evaluate_univariate(rc, challenges[r])

    ref_mls = [x.copy() for x in mls]
    want_coeffs, want_final = oracle.bivariate_sumcheck_prove(ref_mls, n_vars, comps, sums, batch_coeff, challenges)
    assert got_coeffs == want_coeffs
    assert final == want_final
    # final evals == multilinear extensions at the reversed challenges (High-to-Low binding)
    point = list(reversed(challenges))
    for j in range(m):
        assert final[j] == oracle.mle_evaluate(mls[j], n_vars, point)
    # and the last running sum is the batched composition of the final evals
    acc, p = 0, 1
    for i, j in comps:
        acc ^= oracle.mul(p, oracle.mul(final[i], final[j]))
        p = oracle.mul(p, batch_coeff)
    assert acc == running
    # inputs are untouched (PreFold buffers are read-only: first fold copies, bivariate_product.rs:197-205)
    for j in range(m):
        assert np.array_equal(hal.copy_d2h(d[j]), mls[j])


def test_mlecheck_round_evals(hal, oracle):
    """crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:391-520: product * eq_ind as the
    last composition variable, eq table of 2^(n-1) entries shared by both evaluation points."""
    from binius_amd.sumcheck import calculate_round_evals, eq_ind_partial_eval

    alloc = hal.dev_alloc()
    n_vars, m = 9, 3
    mls = [oracle.random_b128(0xB1A50000 + j, 1 << n_vars) for j in range(m)]
    d = [upload(hal, alloc, x) for x in mls]
    point = oracle.random_scalars(0xE9, n_vars - 1)
    eq = eq_ind_partial_eval(hal, alloc, point)
    eq_h = hal.copy_d2h(eq)
    comps = [(0, 1), (2, 0)]
    exprs = [hal.compile_expr([("var", i), ("var", j), ("mul", 0, 1), ("var", m), ("mul", 2, 3)]) for i, j in comps]
    batch_coeff = oracle.random_scalars(0xC4A1, 1)[0]
    coeffs = powers(oracle, batch_coeff, len(comps))
    got = calculate_round_evals(hal, n_vars, coeffs, d, exprs, eq_ind=eq)
    half = 1 << (n_vars - 1)
    want = [0, 0]
    for (i, j), cf in zip(comps, coeffs):
        a, b = mls[i], mls[j]
        p1 = oracle.mul_vec(oracle.mul_vec(a[half:], b[half:]), eq_h)
        pinf = oracle.mul_vec(oracle.mul_vec(a[:half] ^ a[half:], b[:half] ^ b[half:]), eq_h)
        for k, p in enumerate((p1, pinf)):
            s = int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64)
            want[k] ^= oracle.mul(s, cf)
    assert got == want


def test_sumcheck_full_size_properties(oracle):
    """BASELINE config 2 size (n = 24, m = 2): size-independent properties -- the sumcheck verifier's round checks
    and the final product check on the compiled prover's transcript -- plus oracle spot checks of the first round
    (multi-threaded CPU port) and of the first fold."""
    import binius_amd
    from binius_amd._host import SumcheckPlan

    n_vars, m = 24, 2
    n = 1 << n_vars
    hal = binius_amd.Context(0, 3 * n + (1 << 12))
    try:
        alloc = hal.dev_alloc()
        mls = [oracle.random_b128(0xB1A50000 + j, n) for j in range(m)]
        d = [upload(hal, alloc, x) for x in mls]
        s = hal.inner_product(d[0], 7, d[1])
        rc, ev = oracle.round_evals(mls, n_vars, [(0, 1)], 1, threads=8)
        assert rc == 0
        stream = oracle.random_scalars(0xC4A1, n_vars + 1)
        batch_coeff, challenges = stream[0], stream[1:]
        scratch = alloc.alloc(m * (n // 2))
        plan = SumcheckPlan(hal, n_vars, d, scratch, [(0, 1)], [s], batch_coeff, challenges)
        plan.run()
        coeffs, final = plan.round_coeffs(), plan.final_evals()
        running = s
        for r in range(n_vars):
            rcf = coeffs[r]
            if r == 0:
                assert rcf[2] == ev[1] and (running ^ rcf[0]) == ev[0]  # y_inf, y_1 vs the oracle
            assert rcf[0] ^ (rcf[0] ^ rcf[1] ^ rcf[2]) == running
            running = oracle.evaluate_univariate(rcf, challenges[r])
        assert oracle.mul(final[0], final[1]) == running
        # fold spot check at this size: the first 4096 folded elements of each multilinear
        for j in range(m):
            lo, hi = d[j].split_half()
            out = scratch.slice(0, n // 2)
            hal.copy_d2d(lo, out)
            hal.extrapolate_line(out, hi, challenges[0])
            e0 = mls[j][:4096].copy()
            oracle.extrapolate_line(e0, mls[j][n // 2 : n // 2 + 4096], challenges[0])
            assert np.array_equal(hal.copy_d2h(out.slice(0, 4096)), e0)
    finally:
        hal.close()


@pytest.mark.parametrize("n_vars,m,comps", [(1, 2, [(0, 1)]), (9, 2, [(0, 1)]), (13, 3, [(0, 1), (2, 2), (1, 2)]), (17, 2, [(0, 1)])])
def test_compiled_host_prover_matches_oracle(hal, oracle, n_vars, m, comps):
    """The C++ host mirror (binius_amd/host/sumcheck.hpp via libbinius_amd_host.so) drives the same
    C ABI; its round polynomials and final evaluations must equal the oracle's, bit for bit."""
    from binius_amd._host import SumcheckPlan

    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(0xB1A50000 + j, 1 << n_vars) for j in range(m)]
    d = [upload(hal, alloc, x) for x in mls]
    scratch = alloc.alloc(max(1, m * (1 << n_vars) // 2))
    sums = [oracle.inner_product(mls[i], 7, mls[j])[1] for i, j in comps]
    stream = oracle.random_scalars(0xC4A1, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    plan = SumcheckPlan(hal, n_vars, d, scratch, comps, sums, batch_coeff, challenges)
    plan.run()
    want_coeffs, want_final = oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_final
    plan.run()  # re-runnable: inputs are PreFold (never modified)
    assert plan.round_coeffs() == want_coeffs


@pytest.mark.parametrize("n_vars", list(range(2, 13)) + [15, 18, 20])
@pytest.mark.parametrize("out_of_place", [False, True])
def test_fold_then_round_evals_fused(hal, oracle, n_vars, out_of_place):
    """A fold batch followed by the round evaluation of the folded arrays runs as ONE kernel
    (kernels_foldeval9.hip, the ABI defers the fold).  Both the round evaluations and the folded
    arrays left in memory must equal fold-then-evaluate on the oracle.  out_of_place: the first
    fold of the prover (copy evals_0 into a fresh buffer, then fold that) -- the copy is absorbed."""
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    alloc = hal.dev_alloc()
    n = 1 << n_vars
    mls = [oracle.random_b128(0xF01D0000 + j, n) for j in range(2)]
    z = oracle.random_scalars(0xF01D, 1)[0]
    d = [upload(hal, alloc, x) for x in mls]
    halves = [x.split_half() for x in d]
    if out_of_place:
        dst = [alloc.alloc(n // 2) for _ in range(2)]
        for (lo, _), t in zip(halves, dst):
            hal.copy_d2d(lo, t)
    else:
        dst = [lo for lo, _ in halves]
    hal.extrapolate_line_batch(dst, [hi for _, hi in halves], z)
    expr = bivariate_product_expr(hal, 0, 1)
    got = calculate_round_evals(hal, n_vars - 1, [1], dst, [expr])
    folded = []
    for x in mls:
        f = x[: n // 2].copy()
        assert oracle.extrapolate_line(f, x[n // 2 :].copy(), z) == 0
        folded.append(f)
    rc, want = oracle.round_evals(folded, n_vars - 1, [(0, 1)], 1)
    assert rc == 0 and got == want
    for t, f in zip(dst, folded):
        assert np.array_equal(hal.copy_d2h(t), f)
    if out_of_place:  # the sources are untouched
        for x, dd in zip(mls, d):
            assert np.array_equal(hal.copy_d2h(dd), x)


def test_deferred_fold_is_invisible(hal, oracle):
    """The deferral must not be observable: any API call after the fold batch sees folded data."""
    alloc = hal.dev_alloc()
    n = 1 << 10
    mls = [oracle.random_b128(0xF01E0000 + j, n) for j in range(3)]
    z = oracle.random_scalars(0xF01E, 1)[0]
    d = [upload(hal, alloc, x) for x in mls]
    halves = [x.split_half() for x in d]
    hal.extrapolate_line_batch([lo for lo, _ in halves], [hi for _, hi in halves], z)
    for x, (lo, _) in zip(mls, halves):
        f = x[: n // 2].copy()
        oracle.extrapolate_line(f, x[n // 2 :].copy(), z)
        assert np.array_equal(hal.copy_d2h(lo), f)
    # a deferred fold followed by an unrelated kernel launch (different arrays) still happens first
    d2 = [upload(hal, alloc, x) for x in mls[:2]]
    h2 = [x.split_half() for x in d2]
    hal.extrapolate_line_batch([lo for lo, _ in h2], [hi for _, hi in h2], z)
    got = hal.inner_product(h2[0][0], 7, h2[1][0])
    fa, fb = mls[0][: n // 2].copy(), mls[1][: n // 2].copy()
    oracle.extrapolate_line(fa, mls[0][n // 2 :].copy(), z)
    oracle.extrapolate_line(fb, mls[1][n // 2 :].copy(), z)
    assert got == oracle.inner_product(fa, 7, fb)[1]


def _rounds_with_oracle(hal, oracle, n_vars, disturb=None, seed=0x7A110000, disturb_after_fold=None):
    """Drive evaluate -> fold -> evaluate ... through the Python mirror (fold as one batch, the shape
    the ABI fuses and, for small arrays, hands to the resident tail kernel) and check every round's
    (y_1, y_inf) and the final folded values against the oracle.  disturb(round) may poke the
    context between rounds."""
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(seed + j, 1 << n_vars) for j in range(2)]
    zs = oracle.random_scalars(seed ^ 0x55, n_vars)
    d = [upload(hal, alloc, x) for x in mls]
    expr = bivariate_product_expr(hal, 0, 1)
    cur = [x.copy() for x in mls]
    for r in range(n_vars):
        nv = n_vars - r
        got = calculate_round_evals(hal, nv, [1], d, [expr])
        rc, want = oracle.round_evals(cur, nv, [(0, 1)], 1)
        assert rc == 0 and got == want, f"round {r}"
        if disturb is not None:
            disturb(r, d)
        halves = [x.split_half() for x in d]
        hal.extrapolate_line_batch([lo for lo, _ in halves], [hi for _, hi in halves], zs[r])
        if disturb_after_fold is not None:
            disturb_after_fold(r, d)
        nxt = []
        for x in cur:
            f = x[: len(x) // 2].copy()
            assert oracle.extrapolate_line(f, x[len(x) // 2 :].copy(), zs[r]) == 0
            nxt.append(f)
        cur = nxt
        d = [lo for lo, _ in halves]
    for dd, x in zip(d, cur):
        assert np.array_equal(hal.copy_d2h(dd), x)


@pytest.fixture()
def hal_tail(monkeypatch):
    """A context with the resident tail kernel switched on (BN_TAIL_MAX_LOG2, off by default)."""
    import binius_amd

    monkeypatch.setenv("BN_TAIL_MAX_LOG2", "12")
    ctx = binius_amd.Context(0, 1 << 17)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("n_vars", [3, 4, 7, 12, 13, 15])
def test_rounds_through_resident_tail(hal_tail, oracle, n_vars):
    """Rounds with <= 2^12 elements per array run inside ONE resident kernel that is fed the
    challenges through pinned host memory (kernels_foldeval9.hip k_foldeval_tail)."""
    _rounds_with_oracle(hal_tail, oracle, n_vars)


@pytest.mark.parametrize("n_vars", [3, 6, 13, 16])
def test_rounds_without_resident_tail(hal, oracle, n_vars):
    _rounds_with_oracle(hal, oracle, n_vars)


def test_resident_tail_is_cancelled_by_other_calls(hal_tail, oracle):
    hal = hal_tail
    """Any other API call while the tail kernel is parked cancels it; the rounds continue on the
    ordinary kernels (and a new tail may start) with identical results."""
    def disturb(r, d):
        if r in (3, 4, 8):
            hal.copy_d2h(d[0].slice(0, 1))
        if r == 6:
            hal.sync()

    _rounds_with_oracle(hal, oracle, 12, disturb)


def test_resident_tail_times_out_safely(hal_tail, oracle):
    hal = hal_tail
    """A host that stops talking cannot hang the GPU: the parked kernel leaves after a bounded spin
    and the next round falls back to an ordinary launch."""
    import time

    def disturb(r, d):
        if r == 5:
            time.sleep(8.0)

    _rounds_with_oracle(hal, oracle, 9, disturb)


# ---- armed rounds (csrc/arm.hpp): behind the fused kernel of a small round the dispatcher enqueues the kernel of the
# next round, which waits on the device for its challenge.  Every round above already runs that way (the default);
# these tests pin the protocol's edges.


@pytest.fixture(scope="module")
def hal_one_round():
    """A context with the two-round launches switched off (BN_TWO_ROUND=0, read at context creation): every small round is one
    launch of the one-round kernels, which is what the counters below count."""
    import os

    import binius_amd

    old = os.environ.get("BN_TWO_ROUND")
    os.environ["BN_TWO_ROUND"] = "0"
    try:
        ctx = binius_amd.Context(0, 1 << 19)
    finally:
        if old is None:
            os.environ.pop("BN_TWO_ROUND", None)
        else:
            os.environ["BN_TWO_ROUND"] = old
    yield ctx
    ctx.close()


def _needs_arming():
    import os

    if os.environ.get("BN_ARM", "1")[:1] == "0" or os.environ.get("BN_NO_LAZY_FOLD"):
        pytest.skip("armed rounds are switched off in this environment (BN_ARM=0 / BN_NO_LAZY_FOLD)")


@pytest.mark.last
@pytest.mark.parametrize("n_vars", [3, 5, 12, 17])
def test_armed_rounds_serve_the_small_rounds(hal_one_round, oracle, n_vars):
    """Round 0 is a plain evaluation, round 1 the first fused launch; from round 2 on every (small) round is answered by
    a kernel that was already on the device."""
    hal = hal_one_round
    _needs_arming()
    c0 = hal.arm_counters()
    _rounds_with_oracle(hal, oracle, n_vars, seed=0xA4A40000 + n_vars)
    c1 = hal.arm_counters()
    hits, expired = c1["hits"] - c0["hits"], c1["expired"] - c0["expired"]
    # (the oracle's own rounds at 2^14 elements and more take longer than the armed kernel is willing to wait)
    # the invariant is the count: every small round was either answered by the waiting kernel or -- a busy host may miss
    # the 6 ms window -- ran as an ordinary launch after the kernel left; how many of each is the host's timing, not ours
    assert hits + expired == n_vars - 2


@pytest.mark.last
def test_armed_round_is_cancelled_by_other_calls(hal_one_round, oracle):
    """Any call that is not the predicted fold + evaluation pair sends the waiting kernel home (it has touched nothing);
    the round then runs as an ordinary launch and the next one is armed again."""
    hal = hal_one_round
    _needs_arming()
    def after_eval(r, d):
        if r in (2, 3, 7):
            hal.copy_d2h(d[0].slice(0, 1))
        if r == 5:
            hal.sync()

    def after_fold(r, d):
        if r in (4, 8):
            hal.copy_d2h(d[1].slice(0, 1))  # forces the deferred fold out on its own

    c0 = hal.arm_counters()
    _rounds_with_oracle(hal, oracle, 12, after_eval, seed=0xA4A50000, disturb_after_fold=after_fold)
    c1 = hal.arm_counters()
    assert c1["cancels"] - c0["cancels"] >= 5  # (round 5 follows a fold that ran on its own: not fused, so nothing was armed)
    assert 0 < c1["hits"] - c0["hits"] < 10


@pytest.mark.last
def test_armed_round_with_other_arrays_is_cancelled(hal_one_round, oracle):
    """The prediction is 'the same two arrays, in place, half the size'.  Two sumchecks that take turns on one context
    (fold + evaluate of A, then fold + evaluate of B, ...) miss it every time: each fused launch arms a kernel for its
    own next round, the other instance's fold cancels it -- and everybody still gets the right answers."""
    hal = hal_one_round
    _needs_arming()
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    n_vars = 9
    alloc = hal.dev_alloc()
    inst = []
    for k in range(2):
        mls = [oracle.random_b128(0xA4A60000 + 16 * k + j, 1 << n_vars) for j in range(2)]
        inst.append({"cur": [x.copy() for x in mls], "d": [upload(hal, alloc, x) for x in mls]})
    zs = oracle.random_scalars(0xA4A6, 2 * n_vars)
    expr = bivariate_product_expr(hal, 0, 1)
    c0 = hal.arm_counters()
    for r in range(n_vars):
        for k, it in enumerate(inst):
            if r > 0:
                z = zs[2 * (r - 1) + k]
                halves = [x.split_half() for x in it["d"]]
                hal.extrapolate_line_batch([lo for lo, _ in halves], [hi for _, hi in halves], z)
                nxt = []
                for x in it["cur"]:
                    f = x[: len(x) // 2].copy()
                    assert oracle.extrapolate_line(f, x[len(x) // 2 :].copy(), z) == 0
                    nxt.append(f)
                it["cur"] = nxt
                it["d"] = [lo for lo, _ in halves]
            nv = n_vars - r
            got = calculate_round_evals(hal, nv, [1], it["d"], [expr])
            rc, want = oracle.round_evals(it["cur"], nv, [(0, 1)], 1)
            assert rc == 0 and got == want, f"instance {k} round {r}"
    c1 = hal.arm_counters()
    assert c1["hits"] == c0["hits"] and c1["cancels"] - c0["cancels"] >= 2 * (n_vars - 3)


@pytest.mark.last
def test_armed_round_times_out_safely(hal_one_round, oracle):
    """A host that stops talking cannot hang the GPU: the armed kernel leaves after a bounded spin (~6 ms), says so in
    the status word, and the round it was meant for runs as an ordinary launch."""
    hal = hal_one_round
    _needs_arming()
    import time

    def disturb(r, d):
        if r in (3, 6):
            time.sleep(0.5)

    c0 = hal.arm_counters()
    _rounds_with_oracle(hal, oracle, 10, disturb, seed=0xA4A70000)
    c1 = hal.arm_counters()
    assert c1["expired"] - c0["expired"] >= 2


def test_rounds_with_arming_switched_off(oracle, monkeypatch):
    import binius_amd

    monkeypatch.setenv("BN_ARM", "0")
    ctx = binius_amd.Context(0, 1 << 17)
    try:
        _rounds_with_oracle(ctx, oracle, 11, seed=0xA4A80000)
        assert ctx.arm_counters()["hits"] == 0
    finally:
        ctx.close()


def test_tiny_fold_results_are_mirrored_to_the_host(hal, oracle):
    """finish(): the last fold leaves one element per multilinear and the prover reads them back one
    by one; the ABI folds and mirrors them into the pinned mailbox in one launch.  Reads must see
    fresh data also after further folds of the same arrays."""
    alloc = hal.dev_alloc()
    for n in (2, 4, 16, 64, 128):
        mls = [oracle.random_b128(0x71000 + 17 * n + j, n) for j in range(3)]
        d = [upload(hal, alloc, x) for x in mls]
        cur = [x.copy() for x in mls]
        zs = oracle.random_scalars(0x7100 + n, 8)
        r = 0
        while len(cur[0]) > 1:
            halves = [x.split_half() for x in d]
            hal.extrapolate_line_batch([lo for lo, _ in halves], [hi for _, hi in halves], zs[r])
            nxt = []
            for x in cur:
                f = x[: len(x) // 2].copy()
                oracle.extrapolate_line(f, x[len(x) // 2 :].copy(), zs[r])
                nxt.append(f)
            cur, d = nxt, [lo for lo, _ in halves]
            # read back in pieces, twice (second read is served from the same mirror)
            for _ in range(2):
                for dd, x in zip(d, cur):
                    assert np.array_equal(hal.copy_d2h(dd), x)
                    assert np.array_equal(hal.copy_d2h(dd.slice(len(x) - 1, len(x))), x[-1:])
            r += 1


@pytest.mark.parametrize("n_vars,m,comps", [(1, 2, [(0, 1)]), (2, 2, [(0, 1)]), (8, 8, [(0, 1), (2, 5), (7, 7), (3, 4)]), (13, 3, [(0, 1), (2, 0)])])
def test_bivariate_mlecheck_prove(hal, oracle, n_vars, m, comps, monkeypatch):
    """generic_test_bivariate_mlecheck_prove_verify (compute_test_utils bivariate_sumcheck.rs:313-458): the literal
    MLE-check prover mirror (BN_MLECHECK=eager: BivariateMLEcheckProver of binius_amd/host/sumcheck.hpp, the trait-op
    sequence of the reference) over the HIP backend -- round polynomials (degree 3) and final values bit-exact
    against the oracle's restatement of v3/bivariate_mlecheck.rs."""
    from binius_amd._host import MlecheckPlan
    from binius_amd.sumcheck import eq_ind_partial_eval

    monkeypatch.setenv("BN_MLECHECK", "eager")
    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(0x3C3C00 + j, 1 << n_vars) for j in range(m)]
    eq_ch = oracle.random_scalars(0x3C3C00 ^ 0xE9, n_vars)
    full = oracle.arr(1 << n_vars)
    full[0] = (1, 0)
    oracle.tensor_expand(full, 0, eq_ch)
    sums = []
    for i, j in comps:
        p = oracle.mul_vec(oracle.mul_vec(mls[i], mls[j]), full)
        sums.append(int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64))
    d = [upload(hal, alloc, x) for x in mls]
    eq_dev = eq_ind_partial_eval(hal, alloc, eq_ch[: n_vars - 1])
    eq_host = hal.copy_d2h(eq_dev)
    stream = oracle.random_scalars(0xC4A2, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, alloc.alloc(max(1, (m + 1) << max(0, n_vars - 1))), comps, sums, bc, ch)
    plan.run()
    assert plan.last_mode() == 0
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq_host.copy(), eq_ch, comps, sums, bc, ch)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_finals
    for j in range(m):  # PreFold inputs are never modified
        assert np.array_equal(hal.copy_d2h(d[j]), mls[j])


@pytest.mark.parametrize("n_vars,m,comps", [(1, 2, [(0, 1)]), (9, 2, [(0, 1)]), (12, 4, [(0, 1), (2, 3), (1, 1)])])
def test_compiled_mlecheck_prover_matches_oracle(hal, oracle, n_vars, m, comps):
    """The C++ BivariateMLEcheckProver mirror (binius_amd/host/sumcheck.hpp) over the C ABI."""
    from binius_amd._host import MlecheckPlan
    from binius_amd.sumcheck import eq_ind_partial_eval

    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(0x3C3C00 + j, 1 << n_vars) for j in range(m)]
    eq_ch = oracle.random_scalars(0x3C3C00 ^ 0xE9, n_vars)
    full = oracle.arr(1 << n_vars)
    full[0] = (1, 0)
    oracle.tensor_expand(full, 0, eq_ch)
    sums = []
    for i, j in comps:
        p = oracle.mul_vec(oracle.mul_vec(mls[i], mls[j]), full)
        sums.append(int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64))
    d = [upload(hal, alloc, x) for x in mls]
    eq_dev = eq_ind_partial_eval(hal, alloc, eq_ch[: n_vars - 1])
    eq_host = hal.copy_d2h(eq_dev)
    scratch = alloc.alloc(max(1, (m + 1) * (1 << n_vars) // 2))
    stream = oracle.random_scalars(0xC4A2, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, scratch, comps, sums, bc, ch)
    plan.run()
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq_host.copy(), eq_ch, comps, sums, bc, ch)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_finals
    plan.run()  # inputs (PreFold) are never modified: re-runnable
    assert plan.round_coeffs() == want_coeffs


def test_independent_provers_share_the_gpu(oracle):
    """Several contexts driven from their own host threads (tools/bench_concurrent.py): every prover's
    transcript is the oracle's, whatever the interleaving of their launches."""
    import threading

    import binius_amd
    from binius_amd._host import SumcheckPlan

    n_vars, m, provers = 14, 2, 4
    stream = oracle.random_scalars(0xC4A1, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    ctxs, want = [], []
    for p in range(provers):
        mls = [oracle.random_b128(0x5EED00 + 16 * p + j, 1 << n_vars) for j in range(m)]
        rc, claim = oracle.inner_product(mls[0], 7, mls[1])
        assert rc == 0
        hal = binius_amd.Context(0, 4 << n_vars)
        alloc = hal.dev_alloc()
        d = []
        for x in mls:
            s = alloc.alloc(x.shape[0])
            hal.copy_h2d(x, s)
            d.append(s)
        plan = SumcheckPlan(hal, n_vars, d, alloc.alloc(m << (n_vars - 1)), [(0, 1)], [claim], batch_coeff, challenges)
        ctxs.append((hal, plan))
        want.append(oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, [(0, 1)], [claim], batch_coeff, challenges))
    go = threading.Barrier(provers)
    errors = []

    def work(i):
        try:
            go.wait()
            for _ in range(5):
                ctxs[i][1].run()
                assert ctxs[i][1].round_coeffs() == want[i][0]
                assert ctxs[i][1].final_evals() == want[i][1]
        except Exception as e:  # noqa: BLE001 -- reported below
            errors.append((i, repr(e)))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(provers)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for hal, _ in ctxs:
        hal.close()
    assert not errors, errors


# ---- the weighted MLE-check prover (binius_amd/host/sumcheck.hpp WeightedMLEcheckProver) and its one new device op
@pytest.mark.parametrize("log_n", [1, 2, 5, 6, 11, 16, 19])
@pytest.mark.parametrize("mask", [1, 2, 3, 5])
def test_extrapolate_line_batch_scaled(hal, oracle, log_n, mask):
    """bn_extrapolate_line_batch_scaled: the batch fold, then the upper half of every marked array times hi_scale.
    Expected from the oracle's extrapolate_line and an element-wise product."""
    alloc = hal.dev_alloc()
    n, count = 1 << log_n, 3
    x0 = [oracle.random_b128(0x5CA0 + 8 * log_n + i, n) for i in range(count)]
    x1 = [oracle.random_b128(0x5CB0 + 8 * log_n + i, n) for i in range(count)]
    z, hs = oracle.random_scalars(0x5CC0 + mask, 2)
    d0, d1 = [upload(hal, alloc, a) for a in x0], [upload(hal, alloc, b) for b in x1]
    hal.extrapolate_line_batch_scaled(d0, d1, z, mask, hs)
    for i in range(count):
        want = x0[i].copy()
        assert oracle.extrapolate_line(want, x1[i], z) == 0
        if (mask >> i) & 1:
            want[n // 2 :] = oracle.mul_vec(np.ascontiguousarray(want[n // 2 :]), oracle.ints_to_arr([hs] * (n // 2)))
        assert np.array_equal(hal.copy_d2h(d0[i]), want), "array %d" % i
        assert np.array_equal(hal.copy_d2h(d1[i]), x1[i])


def test_extrapolate_line_batch_scaled_validation(hal, oracle):
    from binius_amd import BnError

    alloc = hal.dev_alloc()
    a, b = alloc.alloc(3), alloc.alloc(3)
    with pytest.raises(BnError):  # odd length
        hal.extrapolate_line_batch_scaled([a], [b], 3, 1, 5)
    a, b = alloc.alloc(4), alloc.alloc(4)
    with pytest.raises(BnError):  # mask outside the batch
        hal.extrapolate_line_batch_scaled([a], [b], 3, 2, 5)


def _mlecheck_instance(hal, oracle, alloc, n_vars, m, comps, seed, eq_ch=None):
    from binius_amd.sumcheck import eq_ind_partial_eval

    mls = [oracle.random_b128(seed + j, 1 << n_vars) for j in range(m)]
    if eq_ch is None:
        eq_ch = oracle.random_scalars(seed ^ 0xE9, n_vars)
    full = oracle.arr(1 << n_vars)
    full[0] = (1, 0)
    oracle.tensor_expand(full, 0, eq_ch)
    sums = []
    for i, j in comps:
        p = oracle.mul_vec(oracle.mul_vec(mls[i], mls[j]), full)
        sums.append(int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64))
    d = [upload(hal, alloc, x) for x in mls]
    eq_dev = eq_ind_partial_eval(hal, alloc, eq_ch[: n_vars - 1])
    return mls, d, eq_ch, eq_dev, sums


@pytest.mark.parametrize(
    "n_vars,m,comps",
    [(2, 2, [(0, 1)]), (3, 2, [(1, 0)]), (10, 2, [(0, 1)]), (11, 4, [(0, 1), (2, 1), (2, 3)]), (13, 3, [(0, 1), (2, 0)]), (12, 5, [(0, 1), (3, 4)]),
     (15, 2, [(0, 1)]), (18, 2, [(1, 0)]), (19, 3, [(0, 1), (1, 2)])],
)
def test_weighted_mlecheck_prover_matches_oracle(hal, oracle, n_vars, m, comps):
    """The transcript of the weighted prover is the reference prover's, bit for bit: single compositions (the fused
    fold + evaluation kernels with the scaled fold: 9-lane small / 9-lane / matrix-core by size), several compositions
    sharing multilinears (batch fold + scale pass + generic evaluation), multilinears outside every composition."""
    from binius_amd._host import MlecheckPlan

    alloc = hal.dev_alloc()
    mls, d, eq_ch, eq_dev, sums = _mlecheck_instance(hal, oracle, alloc, n_vars, m, comps, 0x3D3D00 + n_vars)
    eq_host = hal.copy_d2h(eq_dev)
    scratch = alloc.alloc(m * (1 << n_vars))
    stream = oracle.random_scalars(0xC4A3, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, scratch, comps, sums, bc, ch)
    plan.run()
    assert plan.last_mode() == 1, "the weighted prover did not run"
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq_host.copy(), eq_ch, comps, sums, bc, ch)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_finals
    for j in range(m):
        assert np.array_equal(hal.copy_d2h(d[j]), mls[j])  # inputs are never modified
    assert np.array_equal(hal.copy_d2h(eq_dev), eq_host)
    plan.run()
    assert plan.round_coeffs() == want_coeffs and plan.final_evals() == want_finals


@pytest.mark.parametrize("case", ["zeta_zero", "zeta_one", "odd_cycle", "square", "small_scratch", "foreign_table", "eager_env"])
def test_weighted_mlecheck_prover_falls_back(hal, oracle, case, monkeypatch):
    """Where the weighted prover does not apply the literal mirror runs, with the same transcript."""
    from binius_amd._host import MlecheckPlan

    n_vars, m, comps = 9, 3, [(0, 1), (1, 2)]
    eq_ch = oracle.random_scalars(0x3E3E, n_vars)
    if case == "zeta_zero":
        eq_ch[3] = 0
    elif case == "zeta_one":
        eq_ch[0] = 1
    elif case == "odd_cycle":
        comps = [(0, 1), (1, 2), (2, 0)]
    elif case == "square":
        comps = [(0, 1), (2, 2)]
    elif case == "eager_env":
        monkeypatch.setenv("BN_MLECHECK", "eager")
    alloc = hal.dev_alloc()
    mls, d, eq_ch, eq_dev, sums = _mlecheck_instance(hal, oracle, alloc, n_vars, m, comps, 0x3E3E00, eq_ch)
    if case == "foreign_table":  # an entry the spot check reads is not the expansion's: the table is used as given
        t = hal.copy_d2h(eq_dev)
        t[4] ^= 1
        hal.copy_h2d(t, eq_dev)
    eq_host = hal.copy_d2h(eq_dev)
    scratch = alloc.alloc((m + 1) * (1 << n_vars) // 2 if case == "small_scratch" else m * (1 << n_vars))
    stream = oracle.random_scalars(0xC4A4, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, scratch, comps, sums, bc, ch)
    plan.run()
    # (0,1),(1,2) colours 1 weighted, 0 and 2 not: 2^n + 2 * 2^(n-1) = 4 * 2^(n-1) = the small scratch exactly -> still weighted
    assert plan.last_mode() == (1 if case == "small_scratch" else 0)
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq_host.copy(), eq_ch, comps, sums, bc, ch)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_finals


@pytest.mark.parametrize("n_vars,m,comps,eager", [(9, 2, [(0, 1)], False), (12, 3, [(0, 1), (1, 2)], False), (7, 2, [(1, 1)], False), (9, 2, [(0, 1)], True)])
def test_mlecheck_prover_handle_round_by_round(hal, oracle, n_vars, m, comps, eager, monkeypatch):
    """bnh_mlecheck_new / execute / fold / finish: the prover driven the way a transcript drives it -- every challenge
    is handed over only after the round polynomial it depends on has been returned -- with the reference's phase
    errors (ExpectedFold / ExpectedExecution / ExpectedFinish, bivariate_mlecheck.rs:273-372)."""
    from binius_amd import BnError
    from binius_amd._host import MlecheckProver

    if eager:
        monkeypatch.setenv("BN_MLECHECK", "eager")
    alloc = hal.dev_alloc()
    mls, d, eq_ch, eq_dev, sums = _mlecheck_instance(hal, oracle, alloc, n_vars, m, comps, 0x3F3F00 + n_vars)
    eq_host = hal.copy_d2h(eq_dev)
    stream = oracle.random_scalars(0xC4A5, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq_host.copy(), eq_ch, comps, sums, bc, ch)
    prover = MlecheckProver(hal, n_vars, d, eq_dev, eq_ch, alloc.alloc(m << n_vars), comps, sums)
    try:
        assert prover.mode == (0 if eager or comps == [(1, 1)] else 1)
        with pytest.raises(BnError, match="ExpectedExecution"):
            prover.fold(ch[0])
        for r in range(n_vars):
            assert prover.execute(bc) == want_coeffs[r]
            with pytest.raises(BnError, match="ExpectedFold"):
                prover.execute(bc)
            if r == n_vars - 1:
                with pytest.raises(BnError, match="ExpectedFold"):
                    prover.finish()
            prover.fold(ch[r])
        with pytest.raises(BnError, match="ExpectedFinish"):
            prover.fold(ch[0])
        assert prover.finish() == want_finals
    finally:
        prover.close()


@pytest.mark.parametrize("log_n", [6, 12, 15, 19])
@pytest.mark.parametrize("mask", [0, 1, 2, 3])
def test_scaled_fold_followed_by_round_evaluation(hal, oracle, log_n, mask):
    """The deferred (scaled) batch fold fused with the round evaluation that reads the folded arrays -- every mask,
    including both arrays scaled (no fused kernel: fold, scale pass and evaluation run separately) -- gives what the
    operations give one after the other."""
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    alloc = hal.dev_alloc()
    n = 1 << log_n
    x = [oracle.random_b128(0x5D00 + 4 * log_n + j, n) for j in range(2)]
    z, hs = oracle.random_scalars(0x5D80 + mask, 2)
    d = [upload(hal, alloc, v) for v in x]
    lo = [s.slice(0, n // 2) for s in d]
    hi = [s.slice(n // 2, n) for s in d]
    if mask:
        hal.extrapolate_line_batch_scaled(lo, hi, z, mask, hs)
    else:
        hal.extrapolate_line_batch(lo, hi, z)
    expr = bivariate_product_expr(hal, 0, 1)
    got = calculate_round_evals(hal, log_n - 1, [1], lo, [expr])
    want_arrays = []
    for j in range(2):
        w = x[j][: n // 2].copy()
        assert oracle.extrapolate_line(w, x[j][n // 2 :].copy(), z) == 0
        if (mask >> j) & 1:
            w[n // 4 :] = oracle.mul_vec(np.ascontiguousarray(w[n // 4 :]), oracle.ints_to_arr([hs] * (n // 4)))
        want_arrays.append(w)
    rc, want = oracle.round_evals(want_arrays, log_n - 1, [(0, 1)], 1)
    assert rc == 0 and got == want
    for j in range(2):
        assert np.array_equal(hal.copy_d2h(lo[j]), want_arrays[j])
    expr.free()


def test_fp4_matrix_path_on_the_small_shapes():
    """Round evaluations of 2^20 points and more run on the FP4 matrix path (csrc/kernels_roundeval_fp4.hip: E2M1
    operands, f32 counts, element loads through LDS).  At those sizes it is checked by the bench's own bit-exact check and
    by test_gpu_at_size; here the same parity tests once more with the switch at zero (BN_FP4_MIN_LOG2=0: every
    matrix-core round evaluation, ragged sizes included), in a fresh process because the switch is read once."""
    import os
    import subprocess
    import sys

    if os.environ.get("BN_FP4_MIN_LOG2") == "0":
        pytest.skip("already the inner run")
    env = dict(os.environ, BN_FP4_MIN_LOG2="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_sumcheck.py"), os.path.join(root, "tests", "test_gpu_hal.py"),
         os.path.join(root, "tests", "test_gpu_at_size.py"), "-x", "-q", "-m", "gpu",
         "-k", "calculate_round_evals or compiled_host_prover or compiled_sumcheck_plan_n20 or provers_agree_at_2p23 or fast_shape or routed or full_size or inner_product"],
        env=env, cwd=root, capture_output=True, text=True, timeout=1200,
    )
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout
