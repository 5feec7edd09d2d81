"""Parity at the sizes the operations are benchmarked at: every op of
profiles/r0x/ops_*.jsonl against the C oracle on 2^20-element inputs (fri_fold: 2^24), so that the
multi-CU grid-stride and tail paths -- and the matrix-core kernels, which only run from 2^17 points --
are checked, not just the single-workgroup paths the small conformance sizes reach.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOG = 20


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, (1 << 24) + (1 << 22))
    yield ctx
    ctx.close()


def rnd(oracle, seed, n):
    return oracle.random_b128(seed, n)


def upload(hal, alloc, arr):
    d = alloc.alloc(arr.shape[0])
    hal.copy_h2d(arr, d)
    return d


def xor_sum(p):
    return int(np.bitwise_xor.reduce(p[:, 0])) | (int(np.bitwise_xor.reduce(p[:, 1])) << 64)


@pytest.mark.parametrize("level", [0, 3, 4, 5, 6, 7])
def test_inner_product_2p20(hal, oracle, level):
    alloc = hal.dev_alloc()
    n_b = 1 << LOG
    a, b = rnd(oracle, 0x1A0 + level, n_b >> (7 - level)), rnd(oracle, 0x1B0, n_b)
    got = hal.inner_product(upload(hal, alloc, a), level, upload(hal, alloc, b))
    rc, want = oracle.inner_product(a, level, b)
    assert rc == 0 and got == want


@pytest.mark.parametrize("n_b", [2 * 131072, 2 * (131072 + 777), 300002, 1 << 18])
def test_inner_product_ragged_lengths_matrix_core_path(hal, oracle, n_b):
    """F x F sums of products from 2^17 points per half take k_roundeval_mfma<SPLIT>; lengths that are not
    a multiple of the 256-point tile exercise its zero-padded last tile."""
    alloc = hal.dev_alloc()
    a, b = rnd(oracle, 0x1C0, n_b), rnd(oracle, 0x1C1, n_b)
    got = hal.inner_product(upload(hal, alloc, a), 7, upload(hal, alloc, b))
    assert got == xor_sum(oracle.mul_vec(a, b))


@pytest.mark.parametrize("level", [0, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("left", [True, False])
def test_fold_left_right_2p20(hal, oracle, level, left):
    alloc = hal.dev_alloc()
    log_q = 4
    mat = rnd(oracle, 0x2A0 + level, (1 << LOG) >> (7 - level))
    vec = rnd(oracle, 0x2B0, 1 << log_q)
    out_len = 1 << (LOG - log_q)
    dm, dv, do = upload(hal, alloc, mat), upload(hal, alloc, vec), alloc.alloc(out_len)
    exp = oracle.arr(out_len)
    if left:
        hal.fold_left(dm, level, dv, do)
        assert oracle.fold_left(mat, level, vec, exp) == 0
    else:
        hal.fold_right(dm, level, dv, do)
        assert oracle.fold_right(mat, level, vec, exp) == 0
    assert np.array_equal(hal.copy_d2h(do), exp)


def test_fri_fold_log_len_20_log_batch_4(hal, oracle):
    import binius_amd

    alloc = hal.dev_alloc()
    log_len, log_batch, n_fold, tw_level = 20, 4, 4, 5
    log_domain = log_len + 1
    s_ref = oracle.ntt_s_evals(tw_level, log_domain)
    s_dev = binius_amd.ntt_s_evals(tw_level, log_domain)
    assert np.array_equal(s_ref, s_dev)
    data = rnd(oracle, 0x3A0, 1 << (log_len + log_batch))
    challenges = oracle.random_scalars(0x3A1, log_batch + n_fold)
    out_len = 1 << (log_len - n_fold)
    din, dout = upload(hal, alloc, data), alloc.alloc(out_len)
    hal.fri_fold(s_dev, tw_level, log_domain, log_len, log_batch, challenges, din, dout)
    exp = oracle.arr(out_len)
    assert oracle.fri_fold(s_ref, tw_level, log_domain, log_len, log_batch, challenges, data, exp) == 0
    assert np.array_equal(hal.copy_d2h(dout), exp)


def test_compute_composite_product_2p20(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << LOG
    a, b = rnd(oracle, 0x4A0, n), rnd(oracle, 0x4A1, n)
    da, db, do = upload(hal, alloc, a), upload(hal, alloc, b), alloc.alloc(n)
    expr = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])
    hal.compute_composite([da, db], do, expr)
    assert np.array_equal(hal.copy_d2h(do), oracle.mul_vec(a, b))


@pytest.mark.parametrize("n", [458752 + 1, 458752 + 224, 458752 + 225, 458752 + 447, 458752 + 448, 500001, (1 << 20) - 3])
def test_compute_composite_product_ragged_sizes_on_the_two_batch_kernel(hal, oracle, n):
    """More products than wave slots (2048 x 224): kernels_mul9.hip takes two wave-batches per rebuild (k_mul9_dual).  Lengths
    that end inside the first batch of a step, on its boundary, inside the second batch and on a step boundary."""
    alloc = hal.dev_alloc()
    a, b = rnd(oracle, 0x4B0 + (n & 7), n), rnd(oracle, 0x4B8 + (n & 7), n)
    da, db, do = upload(hal, alloc, a), upload(hal, alloc, b), alloc.alloc(n)
    guard = alloc.alloc(64)  # right behind the output: must stay untouched
    hal.fill(guard, 0x5A)
    expr = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])
    hal.compute_composite([da, db], do, expr)
    assert np.array_equal(hal.copy_d2h(do), oracle.mul_vec(a, b))
    g = hal.copy_d2h(guard)
    assert (g[:, 0] == 0x5A).all() and (g[:, 1] == 0).all()


def test_pairwise_product_reduce_2p20(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << LOG
    x = rnd(oracle, 0x5A0, n)
    dx = upload(hal, alloc, x)
    outs = [alloc.alloc(n >> (r + 1)) for r in range(LOG)]
    hal.pairwise_product_reduce(dx, outs)
    exp = [oracle.arr(n >> (r + 1)) for r in range(LOG)]
    assert oracle.pairwise_product_reduce(x, exp) == 0
    for o, e in zip(outs, exp):
        assert np.array_equal(hal.copy_d2h(o), e)


@pytest.mark.parametrize("n_vars", [LOG, LOG + 1])
def test_mlecheck_round_evals_n20(hal, oracle, n_vars):
    """v3/bivariate_mlecheck.rs:391-520 at 2^20: product * eq_ind, eq table of 2^19 entries (the three-factor 9-lane kernel);
    at 2^21 the evaluation has 2^20 points and is routed: the indicator folded into b by two element-wise passes, then the
    bivariate matrix-core kernel (abi_kernels.cpp roundeval_product_routed)."""
    from binius_amd.sumcheck import calculate_round_evals, eq_ind_partial_eval

    alloc = hal.dev_alloc()
    m = 2
    mls = [rnd(oracle, 0xB1A50000 + j, 1 << n_vars) for j in range(m)]
    d = [upload(hal, alloc, x) for x in mls]
    point = oracle.random_scalars(0xE9, n_vars - 1)
    eq = eq_ind_partial_eval(hal, alloc, point)
    eq_h = hal.copy_d2h(eq)
    # the device expansion against the oracle's, and the oracle's against the closed form (test_oracle_pins.py)
    full = oracle.arr(1 << (n_vars - 1))
    full[0] = (1, 0)
    oracle.tensor_expand(full, 0, point)
    assert np.array_equal(eq_h, full)
    expr = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", m), ("mul", 2, 3)])
    coeff = oracle.random_scalars(0xC4A1, 1)[0]
    got = calculate_round_evals(hal, n_vars, [coeff], d, [expr], eq_ind=eq)
    half = 1 << (n_vars - 1)
    a, b = mls
    p1 = oracle.mul_vec(oracle.mul_vec(a[half:], b[half:]), eq_h)
    pinf = oracle.mul_vec(oracle.mul_vec(a[:half] ^ a[half:], b[:half] ^ b[half:]), eq_h)
    assert got == [oracle.mul(xor_sum(p1), coeff), oracle.mul(xor_sum(pinf), coeff)]


def test_compiled_sumcheck_plan_n20(hal, oracle):
    """The compiled prover bench.py times (SumcheckPlan), oracle-checked at 2^20: rounds 0..2 run the
    matrix-core kernels, the rest the 9-lane and small kernels."""
    from binius_amd._host import SumcheckPlan

    alloc = hal.dev_alloc()
    n_vars, m, comps = LOG, 2, [(0, 1)]
    mls = [rnd(oracle, 0xB1A50000 + j, 1 << n_vars) for j in range(m)]
    d = [upload(hal, alloc, x) for x in mls]
    scratch = alloc.alloc(m * (1 << n_vars) // 2)
    sums = [oracle.inner_product(mls[0], 7, mls[1])[1]]
    stream = oracle.random_scalars(0xC4A1, n_vars + 1)
    batch_coeff, challenges = stream[0], stream[1:]
    plan = SumcheckPlan(hal, n_vars, d, scratch, comps, sums, batch_coeff, challenges)
    plan.run()
    want_coeffs, want_final = oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, sums, batch_coeff, challenges, threads=8)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_final
    for j in range(m):
        assert np.array_equal(hal.copy_d2h(d[j]), mls[j])  # PreFold inputs are never modified


def test_compiled_sumcheck_plan_n22_fp4_round0(oracle):
    """From 2^20 points a round evaluation runs on the FP4 matrix path (csrc/kernels_roundeval_fp4.hip); at n = 22 that is
    round 0 (2^21 points) and round 1's inner evaluation of the claim below (2^21 points per half).  The whole transcript
    against the oracle's prover."""
    import binius_amd
    from binius_amd._host import SumcheckPlan

    n_vars, m, comps = 22, 2, [(0, 1)]
    with binius_amd.Context(0, 4 << n_vars) as hal:
        alloc = hal.dev_alloc()
        mls = [rnd(oracle, 0xF4F40000 + j, 1 << n_vars) for j in range(m)]
        d = [upload(hal, alloc, x) for x in mls]
        scratch = alloc.alloc(m * (1 << n_vars) // 2)
        claim = hal.inner_product(d[0], 7, d[1])  # 2^21 points per half: the FP4 kernel's split form
        assert claim == oracle.inner_product(mls[0], 7, mls[1])[1]
        stream = oracle.random_scalars(0xC4A1, n_vars + 1)
        batch_coeff, challenges = stream[0], stream[1:]
        plan = SumcheckPlan(hal, n_vars, d, scratch, comps, [claim], batch_coeff, challenges)
        plan.run()
        want_coeffs, want_final = oracle.bivariate_sumcheck_prove([x.copy() for x in mls], n_vars, comps, [claim], batch_coeff, challenges, threads=8)
        assert plan.round_coeffs() == want_coeffs
        assert plan.final_evals() == want_final


def test_compiled_mlecheck_plan_n20(hal, oracle):
    from binius_amd._host import MlecheckPlan
    from binius_amd.sumcheck import eq_ind_partial_eval

    alloc = hal.dev_alloc()
    n_vars, m, comps = LOG, 2, [(0, 1)]
    mls = [rnd(oracle, 0x3C3C00 + j, 1 << n_vars) for j in range(m)]
    eq_ch = oracle.random_scalars(0x3C3C00 ^ 0xE9, n_vars)
    full = oracle.arr(1 << n_vars)
    full[0] = (1, 0)
    oracle.tensor_expand(full, 0, eq_ch)
    sums = [xor_sum(oracle.mul_vec(oracle.mul_vec(mls[0], mls[1]), full))]
    d = [upload(hal, alloc, x) for x in mls]
    eq_dev = eq_ind_partial_eval(hal, alloc, eq_ch[: n_vars - 1])
    eq_host = hal.copy_d2h(eq_dev)
    scratch = alloc.alloc((m + 1) * (1 << n_vars) // 2)
    stream = oracle.random_scalars(0xC4A2, n_vars + 1)
    bc, ch = stream[0], stream[1:]
    plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, scratch, comps, sums, bc, ch)
    plan.run()
    want_coeffs, want_finals = oracle.bivariate_mlecheck_prove([x.copy() for x in mls], n_vars, eq_host.copy(), eq_ch, comps, sums, bc, ch)
    assert plan.round_coeffs() == want_coeffs
    assert plan.final_evals() == want_finals


# ---- crates/compute_test_utils/src/layer.rs:235-327 test_generic_map_with_multilinear_evaluations
@pytest.mark.parametrize("n_vars", [8, 14, 20])
def test_map_with_multilinear_evaluations(hal, oracle, n_vars):
    """One execute scope: write 1 into element 0 of a zeroed device buffer, tensor_expand it by n_vars
    coordinates, and `map` an inner_product of two multilinears with the expansion.  Expected values come
    from formulas that do not share code with the ops under test: the expansion against the closed form
    prod_j (bit_j(i) ? r_j : 1 - r_j) (spot-checked at 2^20), the evaluations against oracle.mle_evaluate
    (a fold-based evaluator)."""
    alloc = hal.dev_alloc()
    n = 1 << n_vars
    mle1, mle2 = rnd(oracle, 0x6A0, n), rnd(oracle, 0x6A1, n)
    d1, d2 = upload(hal, alloc, mle1), upload(hal, alloc, mle2)
    eq = alloc.alloc(n)
    hal.copy_h2d(oracle.arr(n), eq)
    coords = oracle.random_scalars(0x6A2, n_vars)
    hal.copy_h2d(oracle.ints_to_arr([1]), eq.slice(0, 1))
    hal.tensor_expand(0, coords, eq)
    evals = [hal.inner_product(d, 7, eq) for d in (d1, d2)]  # ComputeLayerExecutor::map over the two multilinears
    got_eq = hal.copy_d2h(eq)
    idx = range(n) if n_vars <= 8 else [0, 1, 2, n // 3, n // 2 + 5, n - 2, n - 1]
    for i in idx:
        want = 1
        for j, r in enumerate(coords):
            want = oracle.mul(want, r if (i >> j) & 1 else (1 ^ r))
        assert (int(got_eq[i, 0]) | (int(got_eq[i, 1]) << 64)) == want
    assert evals[0] == oracle.mle_evaluate(mle1, n_vars, coords)
    assert evals[1] == oracle.mle_evaluate(mle2, n_vars, coords)


def test_weighted_and_literal_mlecheck_provers_agree_at_2p23(oracle, monkeypatch):
    """n = 23, m = 2: the weighted prover (matrix-core rounds on the indicator-weighted factor) and the literal
    trait-call sequence produce the same transcript, and the transcript passes the MLE-check verifier's round
    equations (v3/bivariate_mlecheck.rs:273-318 read backwards: the prime polynomial's value at the indicator
    coordinate chains from round to round)."""
    import binius_amd
    from binius_amd import synthetic
    from binius_amd._host import MlecheckPlan
    from binius_amd.sumcheck import eq_ind_partial_eval

    n_vars, m = 23, 2
    n = 1 << n_vars
    hal = binius_amd.Context(0, 2 * n + n // 2 + 3 * n // 2 + (1 << 12))
    try:
        alloc = hal.dev_alloc()
        d = []
        for j in range(m):
            s = alloc.alloc(n)
            hal.copy_h2d(synthetic.random_b128(0x7A70 + j, n), s)
            d.append(s)
        eq_ch = synthetic.random_scalars(0x7A7E, n_vars)
        eq_dev = eq_ind_partial_eval(hal, alloc, eq_ch[: n_vars - 1])
        scratch = alloc.alloc(3 * n // 2)
        stream = synthetic.random_scalars(0x7A7F, n_vars + 1)
        bc, ch = stream[0], stream[1:]
        # the claimed sum: sum_x eq(x) a(x) b(x) with the full indicator = both halves of the table times (1 - z), z
        top = eq_ch[n_vars - 1]
        lo_sum = hal.hal_round_evals(1, n_vars, None, [("folded", d[0], 0), ("folded", d[1], 0)],
                                     [{"composition": (e := hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])), "composition_at_infinity": e,
                                       "start": 1, "end": 2, "eq_ind": eq_dev}], [])[0][0]
        # (only the X = 1 half is needed for the check below; the claim itself is an input of the protocol)
        plan = MlecheckPlan(hal, n_vars, d, eq_dev, eq_ch, scratch, [(0, 1)], [0], bc, ch)
        plan.run()
        assert plan.last_mode() == 1
        weighted = (plan.round_coeffs(), plan.final_evals())
        monkeypatch.setenv("BN_MLECHECK", "eager")
        plan.run()
        assert plan.last_mode() == 0
        assert (plan.round_coeffs(), plan.final_evals()) == weighted
        # round 0: the degree-3 polynomial v(X) = v'(X) * eq(X, top); v(1) = y_1 * top with y_1 = sum over the upper half
        c = weighted[0][0]
        v1 = c[0] ^ c[1] ^ c[2] ^ c[3]
        assert v1 == oracle.mul(lo_sum, top)
        e.free()
    finally:
        hal.close()


@pytest.mark.parametrize("grid", [100, 248])
def test_fused_fp4_kernel_with_odd_tile_counts(grid):
    """kernels_foldeval_fp4.hip and the wave-specialised k_roundeval_fp4_ws hand PAIRS of tiles from their fold / stager waves to their Gram waves.  On 256 CUs every size a prover
    reaches gives every workgroup an even number of tiles and the XCD-aware tile order, so the odd last pair (fold group 1
    idle, the Gram waves run four k-steps instead of eight) and the plain striding never run.  BN_FE_FP4_GRID / BN_FP4_WS_GRID launch the
    kernels on another grid -- 100 workgroups: plain striding, 248 = 8 x 31: the XCD-aware order with uneven counts -- and the
    transcripts of the n = 20 / 22 plans, the 2^23 MLE-check provers (scaled folds) and the n = 20 round evaluations must still
    be the oracle's.  A fresh process, because the switch is read once."""
    import os
    import subprocess
    import sys

    if os.environ.get("BN_FE_FP4_GRID"):
        pytest.skip("already the inner run")
    env = dict(os.environ, BN_FE_FP4_GRID=str(grid), BN_FP4_WS_GRID=str(grid))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_at_size.py"), "-x", "-q", "-m", "gpu",
         "-k", "compiled_sumcheck_plan or compiled_mlecheck_plan or provers_agree_at_2p23 or mlecheck_round_evals"],
        env=env, cwd=root, capture_output=True, text=True, timeout=1200,
    )
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout
