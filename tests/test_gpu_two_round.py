"""Two sumcheck rounds per launch for the small rounds (csrc/kernels_foldeval8.hip, abi_kernels.cpp).

A launch of the two-round kernel answers its own round AND leaves the next round's evaluations as quadratics in the next
challenge, so that the host answers the next `accumulate_kernels` without a launch and the launch after that folds twice.
It is an execution detail of the unchanged call sequence of the v3 prover (execute -> fold -> execute ...,
crates/core/src/protocols/sumcheck/v3/bivariate_product.rs:133-232): every value, every array in memory and every
interleaving with other calls must be what the one-round kernels (BN_TWO_ROUND=0) and the oracle produce.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, 1 << 21)
    yield ctx
    ctx.close()


def upload(hal, alloc, arr):
    d = alloc.alloc(arr.shape[0])
    hal.copy_h2d(arr, d)
    return d


def _drive(hal, oracle, n_vars, seed, after_eval=None, after_fold=None, model=None):
    """evaluate -> fold -> evaluate ... through the Python mirror of the trait calls, in place, every round against the
    oracle.  model: a list of host copies of the WHOLE buffers, folded in place like the device does, for memory checks."""
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    alloc = hal.dev_alloc()
    mls = [oracle.random_b128(seed + j, 1 << n_vars) for j in range(2)]
    zs = oracle.random_scalars(seed ^ 0x55, n_vars)
    full = [upload(hal, alloc, x) for x in mls]
    d = list(full)
    expr = bivariate_product_expr(hal, 0, 1)
    cur = [x.copy() for x in mls]
    if model is not None:
        model.extend(x.copy() for x in mls)
    for r in range(n_vars):
        nv = n_vars - r
        got = calculate_round_evals(hal, nv, [1], d, [expr])
        rc, want = oracle.round_evals(cur, nv, [(0, 1)], 1)
        assert rc == 0 and got == want, f"round {r}"
        if after_eval is not None:
            after_eval(r, d, full)
        halves = [x.split_half() for x in d]
        hal.extrapolate_line_batch([lo for lo, _ in halves], [hi for _, hi in halves], zs[r])
        nxt = []
        for j, x in enumerate(cur):
            f = x[: len(x) // 2].copy()
            assert oracle.extrapolate_line(f, x[len(x) // 2 :].copy(), zs[r]) == 0
            nxt.append(f)
            if model is not None:
                model[j][: len(f)] = f
        cur = nxt
        d = [lo for lo, _ in halves]
        if after_fold is not None:
            after_fold(r, d, full)
    for dd, x in zip(d, cur):
        assert np.array_equal(hal.copy_d2h(dd), x)
    return full


@pytest.fixture(scope="module")
def hal_no_host_tail():
    """BN_HOST_TAIL=0 (read at context creation): the chain of two-round launches runs down to four elements."""
    import os

    import binius_amd

    old = os.environ.get("BN_HOST_TAIL")
    os.environ["BN_HOST_TAIL"] = "0"
    try:
        ctx = binius_amd.Context(0, 1 << 21)
    finally:
        if old is None:
            os.environ.pop("BN_HOST_TAIL", None)
        else:
            os.environ["BN_HOST_TAIL"] = old
    yield ctx
    ctx.close()


@pytest.mark.parametrize("n_vars", [2, 3, 4, 5, 6, 7, 8, 11, 12, 15, 16, 17])
def test_two_round_launches_answer_two_rounds_each(hal_no_host_tail, oracle, n_vars):
    """Counted, not timed: with an even number of variables round 0 is itself a two-round launch (no fold), with an odd
    number it runs alone; from then on every launch covers two rounds and the chain ends on four elements."""
    hal = hal_no_host_tail
    c0 = hal.arm_counters()
    _drive(hal, oracle, n_vars, 0x2B2B0000 + 64 * n_vars)
    c1 = hal.arm_counters()
    launches, hosted = c1["two_round"] - c0["two_round"], c1["hosted"] - c0["hosted"]
    small = min(n_vars - n_vars % 2, 16)  # rounds inside the two-round regime: Y of at most 2^16 elements, an even exponent
    assert launches == small // 2 and hosted == small // 2, (launches, hosted)
    assert c1["ht_started"] == c0["ht_started"] and c1["ht_rounds"] == c0["ht_rounds"]


@pytest.mark.parametrize("n_vars", [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 16, 17])
def test_host_tail_takes_the_last_rounds(hal, oracle, n_vars):
    """The default: the first two-round launch whose Y has at most 2^12 (without VPCLMULQDQ on the host: 2^8) elements per array hands Y to the host (in the host's
    power basis), and every later evaluation and fold of the instance is host arithmetic -- no launch until the caller reads
    the final evaluations, when ONE launch performs all the outstanding folds on the device (csrc/abi_kernels.cpp "host
    tail").  Counted; every round's values against the oracle inside _drive, the final read included."""
    c0 = hal.arm_counters()
    _drive(hal, oracle, n_vars, 0x2B3B0000 + 64 * n_vars)
    c1 = hal.arm_counters()
    d = {k: c1[k] - c0[k] for k in c1}
    ht_log = c1["ht_max"].bit_length() - 1        # 12 when the host folds on VPCLMULQDQ, else 8 (BN_HOST_TAIL_MAX_LOG2 moves it)
    assert ht_log >= 2 and ht_log % 2 == 0
    small = min(n_vars - n_vars % 2, 16)          # exponent of the first Y of the chain (even)
    take = min(small, ht_log)                     # exponent of the Y the host takes over
    launches = (small - take) // 2 + 1            # two-round launches: Y of 2^small, 2^(small-2), ..., 2^take
    assert d["two_round"] == launches and d["hosted"] == launches - 1, d
    assert d["ht_started"] == 1 and d["ht_flushed"] == 1, d
    assert d["ht_rounds"] == take - 1, d          # the rounds on 2^(take-1), ..., 2^1 elements


@pytest.mark.parametrize("which", ["default", "no_host_tail"])
def test_two_round_path_leaves_memory_as_two_separate_folds_would(hal, hal_no_host_tail, oracle, which):
    """After the second fold of a pair the caller's buffers hold what two in-place folds leave: Y in the first quarter,
    the upper half of the once-folded array behind it, the rest untouched.  Reading the whole buffer back at different
    points of the pair (which flushes whatever is deferred) must always show exactly that.  With the host tail (the default)
    the same holds for the folds the host performed on its own copy: a read launches the chain of outstanding folds."""
    hal = hal if which == "default" else hal_no_host_tail
    for read_at in ("after_every_fold", "after_even_folds", "after_odd_folds", "after_eval"):
        model = []

        def check(r, d, full, read_at=read_at, model=model):
            if read_at == "after_even_folds" and r % 2:
                return
            if read_at == "after_odd_folds" and r % 2 == 0:
                return
            for j, buf in enumerate(full):
                assert np.array_equal(hal.copy_d2h(buf), model[j]), (read_at, r, j)

        if read_at == "after_eval":
            _drive(hal, oracle, 9, 0x2B2C0000, after_eval=lambda r, d, full: check(r, d, full) if r else None, model=model)
        else:
            _drive(hal, oracle, 10, 0x2B2C0100, after_fold=check, model=model)


@pytest.mark.parametrize("which", ["default", "no_host_tail"])
def test_two_round_path_survives_foreign_calls(hal, hal_no_host_tail, oracle, which):
    """A call that is not the predicted one -- between the evaluation and its fold, between the host-answered round and the
    second fold, or after it -- flushes what is deferred and drops the precomputed sums; the rounds go on with the right
    answers (the one-round kernels take over until the next two-round launch)."""
    hal = hal if which == "default" else hal_no_host_tail

    def after_eval(r, d, full):
        if r in (1, 2, 6, 9):
            hal.copy_d2h(d[0].slice(0, 1))
        if r == 4:
            hal.sync()

    def after_fold(r, d, full):
        if r in (3, 5, 6, 10):
            hal.copy_d2h(d[1].slice(0, 1))

    _drive(hal, oracle, 13, 0x2B2D0000, after_eval=after_eval, after_fold=after_fold)
    _drive(hal, oracle, 12, 0x2B2D0100, after_eval=after_eval)
    _drive(hal, oracle, 12, 0x2B2D0200, after_fold=after_fold)


def test_two_interleaved_sumchecks_on_one_context(hal, oracle):
    """Two provers take turns on one context: every fold of one instance arrives while the other instance's sums (and
    its armed kernel) are what the context remembers.  Everybody still gets the right answers."""
    from binius_amd.sumcheck import bivariate_product_expr, calculate_round_evals

    n_vars = 8
    alloc = hal.dev_alloc()
    inst = []
    for k in range(2):
        mls = [oracle.random_b128(0x2B2E0000 + 16 * k + j, 1 << n_vars) for j in range(2)]
        inst.append({"cur": [x.copy() for x in mls], "d": [upload(hal, alloc, x) for x in mls]})
    zs = oracle.random_scalars(0x2B2E, 2 * n_vars)
    expr = bivariate_product_expr(hal, 0, 1)
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


@pytest.mark.parametrize("n_vars,m,comps", [(2, 2, [(0, 1)]), (3, 2, [(0, 1)]), (8, 2, [(0, 1)]), (9, 2, [(0, 1)]), (16, 2, [(1, 0)]), (17, 2, [(0, 1)]), (19, 2, [(0, 1)]),
                                            (10, 3, [(0, 1), (2, 0)]), (6, 2, [(1, 1)])])
def test_compiled_prover_same_transcript_with_and_without_two_round_launches(oracle, monkeypatch, n_vars, m, comps):
    """The compiled prover bench.py times (first fold out of place into scratch, later folds in place, finish() reading the
    last elements back): identical transcripts with BN_TWO_ROUND=0, with the default, and from the oracle.  Claims with
    several batched compositions or a squared multilinear are not the single-pair shape and keep the one-round kernels."""
    import binius_amd
    from binius_amd._host import SumcheckPlan

    mls = [oracle.random_b128(0x2B2F0000 + 32 * n_vars + j, 1 << n_vars) for j in range(m)]
    sums = [oracle.inner_product(mls[i], 7, mls[j])[1] for i, j in comps]
    stream = oracle.random_scalars(0xC4A1 + n_vars, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    want = oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges, threads=4)
    seen = {}
    for mode in ("0", "1", "1-no-host-tail"):
        monkeypatch.setenv("BN_TWO_ROUND", mode[0])
        monkeypatch.setenv("BN_HOST_TAIL", "0" if mode.endswith("no-host-tail") else "1")
        with binius_amd.Context(0, 4 * m << n_vars) as ctx:
            alloc = ctx.dev_alloc()
            d = [upload(ctx, alloc, x) for x in mls]
            scratch = alloc.alloc(m << max(n_vars - 1, 0))
            plan = SumcheckPlan(ctx, n_vars, d, scratch, comps, sums, batch_coeff, challenges)
            for _ in range(2):  # (the second run starts from the state the first one left)
                plan.run()
                assert plan.round_coeffs() == want[0], "BN_TWO_ROUND=%s" % mode
                assert plan.final_evals() == want[1], "BN_TWO_ROUND=%s" % mode
            seen[mode] = ctx.arm_counters()
            for j in range(m):
                assert np.array_equal(ctx.copy_d2h(d[j]), mls[j])  # PreFold inputs are never modified
    assert seen["0"]["two_round"] == 0 and seen["0"]["hosted"] == 0
    assert seen["0"]["ht_started"] == 0 and seen["1-no-host-tail"]["ht_started"] == 0
    if len(comps) == 1 and comps[0][0] != comps[0][1] and n_vars >= 2:
        assert seen["1-no-host-tail"]["two_round"] > 0 and seen["1-no-host-tail"]["hosted"] == seen["1-no-host-tail"]["two_round"]
        # the default: both runs of the plan were taken over by the host once, each caught up with one launch
        assert seen["1"]["ht_started"] == 2 and seen["1"]["ht_flushed"] == 2 and seen["1"]["two_round"] > 0
    else:
        assert seen["1"]["ht_started"] == 0


@pytest.mark.parametrize("ht_log,vector", [(4, "1"), (6, "1"), (8, "0"), (10, "1"), (10, "0"), (12, "1")])
def test_host_tail_at_other_hand_over_sizes(oracle, monkeypatch, ht_log, vector):
    """The hand-over size is a knob (BN_HOST_TAIL_MAX_LOG2): 2^10 and 2^12 elements are handed over by 4 and 16 workgroups (the
    staging's tag is accumulated on the device and published by the last one), 2^4 by a nearly empty one; the host folds and sums
    on VPCLMULQDQ when it has it (BN_HOSTMUL_VECTOR=0: the scalar PCLMULQDQ loops).  Same values, same memory, whatever the setting."""
    import binius_amd

    monkeypatch.setenv("BN_HOST_TAIL_MAX_LOG2", str(ht_log))
    monkeypatch.setenv("BN_HOSTMUL_VECTOR", vector)
    with binius_amd.Context(0, 1 << 21) as ctx:
        assert ctx.arm_counters()["ht_max"] == 1 << ht_log
        for n_vars in (5, 10, 13, 14, 17):
            model = []

            def check(r, d, full, model=model):
                if r in (n_vars - 3, n_vars - 6):  # a read in the middle of the host rounds: the write-back must leave eager memory
                    for j, buf in enumerate(full):
                        assert np.array_equal(ctx.copy_d2h(buf), model[j]), (ht_log, n_vars, r, j)

            c0 = ctx.arm_counters()
            _drive(ctx, oracle, n_vars, 0x2B4B0000 + 1024 * ht_log + n_vars, after_fold=check, model=model)
            c1 = ctx.arm_counters()
            assert c1["ht_started"] > c0["ht_started"], (ht_log, n_vars)


@pytest.mark.parametrize("n_vars", [9, 12, 14])
def test_deferred_copies_and_the_host_tail_keep_their_order(hal, oracle, n_vars):
    """A copy_d2d OUT of the arrays between an evaluation and their fold is deferred (it may be the "copy evals_0 into a fresh
    buffer" of a first fold); when the fold that follows is performed on the host copy, the copy must still read what the arrays
    held BEFORE the fold -- not what the host tail's write-back leaves there later (found by the differential fuzz at the 2^12
    hand-over: the write-back used to be enqueued ahead of the unabsorbed copies)."""
    alloc_spare = hal.dev_alloc()
    alloc_spare.alloc(4 << n_vars)  # (every dev_alloc() starts at the arena's base: stay clear of the arrays _drive allocates)
    spare = alloc_spare.alloc(1 << n_vars)
    expect = {}

    def after_eval(r, d, full):
        if r >= 1 and r % 2 == 1 and d[0].len >= 4:
            half = d[0].len // 2
            hal.copy_d2d(d[0].slice(0, half), spare.slice(0, half))
            expect["n"] = half
            expect["want"] = hal_model[0][:half].copy()

    def after_fold(r, d, full):
        if "want" in expect:
            got = hal.copy_d2h(spare.slice(0, expect["n"]))
            assert np.array_equal(got, expect.pop("want")), r

    hal_model = []
    # (_drive folds `model` in place like the device: model[0][:len] is array 0 before the fold when after_eval runs)
    _drive(hal, oracle, n_vars, 0x2B5B0000 + n_vars, after_eval=after_eval, after_fold=after_fold, model=hal_model)
