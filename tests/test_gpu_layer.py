"""GPU parity tests: every ComputeLayer op through the C ABI vs the CPU oracle, bit-exact.

These mirror the reference's backend-conformance tests (generic functions in
crates/compute_test_utils/src/layer.rs instantiated per backend in crates/compute/tests/layer.rs):
fill host buffers from a seeded PRNG, copy_h2d, run the op, copy_d2h, assert_eq against an
independent CPU formula.  The reference's StdRng stream cannot be regenerated without the Rust
crate, so inputs come from the documented SplitMix64 streams.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, 1 << 22)
    yield ctx
    ctx.close()


def rnd(oracle, seed, n):
    return oracle.random_b128(seed, n)


def to_int(a, i=0):
    return int(a[i, 0]) | (int(a[i, 1]) << 64)


def upload(hal, alloc, arr):
    d = alloc.alloc(arr.shape[0])
    hal.copy_h2d(arr, d)
    return d


# ---- crates/compute/src/layer.rs:792-825 test_copy_host_device
def test_copy_host_device(hal, oracle):
    alloc = hal.dev_alloc()
    h1 = rnd(oracle, 1, 128)
    d1 = upload(hal, alloc, h1)
    d2 = alloc.alloc(128)
    hal.copy_d2d(d1, d2)
    h2 = hal.copy_d2h(d2)
    assert np.array_equal(h1, h2)


def test_copy_length_mismatch_is_input_validation(hal, oracle):
    import binius_amd

    alloc = hal.dev_alloc()
    d = alloc.alloc(16)
    with pytest.raises(binius_amd.BnError) as e:
        hal.copy_h2d(rnd(oracle, 2, 8), d)
    assert e.value.kind == "InputValidation"
    with pytest.raises(binius_amd.BnError) as e:
        hal.copy_d2d(d, alloc.alloc(8))
    assert e.value.kind == "InputValidation"


def test_bump_allocator(hal):
    """crates/compute/src/alloc.rs:123-158"""
    import binius_amd

    bump = binius_amd.BumpAllocator(hal.arena.slice(0, 256))
    assert bump.alloc(100).len == 100
    assert bump.alloc(100).len == 100
    with pytest.raises(binius_amd.BnError) as e:
        bump.alloc(100)
    assert e.value.kind == "Alloc"
    bump = binius_amd.BumpAllocator(hal.arena.slice(0, 256))
    assert bump.alloc(100).len == 100
    sub = bump.subscope_allocator()
    sub.alloc(100)
    with pytest.raises(binius_amd.BnError):
        sub.alloc(57)
    sub.alloc(56)
    bump.alloc(100)


def test_fill(hal, oracle):
    alloc = hal.dev_alloc()
    d = alloc.alloc(1000)
    v = 0x0123456789ABCDEF_FEDCBA9876543210
    hal.fill(d, v)
    h = hal.copy_d2h(d)
    assert all(to_int(h, i) == v for i in range(1000))


# ---- test_extrapolate_line (compute_test_utils/src/layer.rs:726-771)
@pytest.mark.parametrize("n", [1, 2, 64, 1 << 10, 3000, 1 << 16, (1 << 17) + 77])
def test_extrapolate_line(hal, oracle, n):
    alloc = hal.dev_alloc()
    e0, e1 = rnd(oracle, 10, n), rnd(oracle, 11, n)
    z = oracle.random_scalars(12, 1)[0]
    d0, d1 = upload(hal, alloc, e0), upload(hal, alloc, e1)
    hal.extrapolate_line(d0, d1, z)
    got = hal.copy_d2h(d0)
    exp = e0.copy()
    assert oracle.extrapolate_line(exp, e1, z) == 0
    assert np.array_equal(got, exp)
    assert np.array_equal(hal.copy_d2h(d1), e1)  # evals_1 untouched


def test_extrapolate_line_special_z(hal, oracle):
    alloc = hal.dev_alloc()
    n = 512
    e0, e1 = rnd(oracle, 13, n), rnd(oracle, 14, n)
    for z in (0, 1, 1 << 127, (1 << 128) - 1, 2):
        d0, d1 = upload(hal, alloc, e0), upload(hal, alloc, e1)
        hal.extrapolate_line(d0, d1, z)
        exp = e0.copy()
        oracle.extrapolate_line(exp, e1, z)
        assert np.array_equal(hal.copy_d2h(d0), exp), hex(z)
    # z = 0 keeps evals_0, z = 1 yields evals_1
    d0, d1 = upload(hal, alloc, e0), upload(hal, alloc, e1)
    hal.extrapolate_line(d0, d1, 1)
    assert np.array_equal(hal.copy_d2h(d0), e1)


def test_extrapolate_line_length_mismatch(hal, oracle):
    import binius_amd

    alloc = hal.dev_alloc()
    with pytest.raises(binius_amd.BnError) as e:
        hal.extrapolate_line(alloc.alloc(8), alloc.alloc(4), 3)
    assert e.value.kind == "InputValidation"


# ---- test_generic_single_tensor_expand (layer.rs:22-72) + eq_ind_partial_eval (ops.rs:26-50)
@pytest.mark.parametrize("log_n,k", [(2, 6), (0, 8), (0, 1), (3, 0), (0, 14), (0, 15), (5, 13), (13, 4), (12, 1), (11, 2), (0, 20), (3, 16)])
def test_tensor_expand(hal, oracle, log_n, k):
    alloc = hal.dev_alloc()
    n = 1 << (log_n + k)
    buf = oracle.arr(n)
    buf[: 1 << log_n] = rnd(oracle, 20, 1 << log_n)
    coords = oracle.random_scalars(21, k)
    d = upload(hal, alloc, buf)
    hal.tensor_expand(log_n, coords, d)
    exp = buf.copy()
    assert oracle.tensor_expand(exp, log_n, coords) == 0
    assert np.array_equal(hal.copy_d2h(d), exp)


def test_tensor_expand_overwrites_upper_half(hal, oracle):
    """Documented semantics (layer.rs:282-286, FastCpuLayer): the upper part is an output only."""
    alloc = hal.dev_alloc()
    garbage = rnd(oracle, 22, 64)
    garbage[0] = (1, 0)
    d = upload(hal, alloc, garbage)
    coords = oracle.random_scalars(23, 6)
    hal.tensor_expand(0, coords, d)
    exp = oracle.arr(64)
    exp[0] = (1, 0)
    oracle.tensor_expand(exp, 0, coords)
    assert np.array_equal(hal.copy_d2h(d), exp)


def test_tensor_expand_bad_length(hal):
    import binius_amd

    alloc = hal.dev_alloc()
    with pytest.raises(binius_amd.BnError) as e:
        hal.tensor_expand(2, [1, 2, 3], alloc.alloc(16))
    assert e.value.kind == "InputValidation"


def test_eq_ind_partial_eval_matches_mle(hal, oracle):
    """test_generic_multiple_multilinear_evaluations (layer.rs:127-233): eq table == query expansion
    and inner_product(evals, eq) == MLE evaluation."""
    from binius_amd.sumcheck import eq_ind_partial_eval

    alloc = hal.dev_alloc()
    n_vars = 8
    point = oracle.random_scalars(30, n_vars)
    eq = eq_ind_partial_eval(hal, alloc, point)
    exp = oracle.arr(1 << n_vars)
    exp[0] = (1, 0)
    oracle.tensor_expand(exp, 0, point)
    assert np.array_equal(hal.copy_d2h(eq), exp)
    for level, seed in ((4, 31), (5, 32), (7, 33)):
        a = rnd(oracle, seed, (1 << n_vars) >> (7 - level))
        da = upload(hal, alloc, a)
        got = hal.inner_product(da, level, eq)
        rc, want = oracle.inner_product(a, level, exp)
        assert rc == 0 and got == want
        if level == 7:
            assert got == oracle.mle_evaluate(a, n_vars, point)


# ---- test_generic_single_inner_product (layer.rs:74-125)
@pytest.mark.parametrize("level", [0, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("n_b", [1 << 8, 1 << 13])
def test_inner_product(hal, oracle, level, n_b):
    alloc = hal.dev_alloc()
    n_a = n_b >> (7 - level)
    a, b = rnd(oracle, 40 + level, n_a), rnd(oracle, 50, n_b)
    da, db = upload(hal, alloc, a), upload(hal, alloc, b)
    got = hal.inner_product(da, level, db)
    rc, want = oracle.inner_product(a, level, b)
    assert rc == 0 and got == want


def test_inner_product_validation(hal):
    import binius_amd

    alloc = hal.dev_alloc()
    for level, na, nb in ((4, 8, 32), (8, 8, 8), (2, 8, 8 << 5)):
        with pytest.raises(binius_amd.BnError) as e:
            hal.inner_product(alloc.alloc(na), level, alloc.alloc(nb))
        assert e.value.kind == "InputValidation"


# ---- test_generic_single_left_fold / right_fold (layer.rs:572-724)
@pytest.mark.parametrize("level", [0, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("log_q", [1, 3])
@pytest.mark.parametrize("left", [True, False])
def test_fold_left_right(hal, oracle, level, log_q, left):
    alloc = hal.dev_alloc()
    log_evals = 10
    mat = rnd(oracle, 60 + level, (1 << log_evals) >> (7 - level))
    vec = rnd(oracle, 61, 1 << log_q)
    out_len = 1 << (log_evals - log_q)
    dm, dv, do = upload(hal, alloc, mat), upload(hal, alloc, vec), alloc.alloc(out_len)
    exp = oracle.arr(out_len)
    if left:
        hal.fold_left(dm, level, dv, do)
        assert oracle.fold_left(mat, level, vec, exp) == 0
    else:
        hal.fold_right(dm, level, dv, do)
        assert oracle.fold_right(mat, level, vec, exp) == 0
    assert np.array_equal(hal.copy_d2h(do), exp)


@pytest.mark.parametrize("level,log_q", [(3, 2), (4, 5), (5, 6), (5, 1)])
@pytest.mark.parametrize("left", [True, False])
def test_fold_left_right_table_path(hal, oracle, level, log_q, left):
    """Larger outputs take the LDS nibble-table kernel (out_len >= 1024, levels 3..5)."""
    alloc = hal.dev_alloc()
    log_evals = 13 + log_q - 2
    mat = rnd(oracle, 160 + level, (1 << log_evals) >> (7 - level))
    vec = rnd(oracle, 161, 1 << log_q)
    out_len = 1 << (log_evals - log_q)
    assert out_len >= 1024
    dm, dv, do = upload(hal, alloc, mat), upload(hal, alloc, vec), alloc.alloc(out_len)
    exp = oracle.arr(out_len)
    if left:
        hal.fold_left(dm, level, dv, do)
        assert oracle.fold_left(mat, level, vec, exp) == 0
    else:
        hal.fold_right(dm, level, dv, do)
        assert oracle.fold_right(mat, level, vec, exp) == 0
    assert np.array_equal(hal.copy_d2h(do), exp)


@pytest.mark.parametrize("level,log_q", [(0, 9), (0, 10), (0, 11), (3, 6), (3, 7), (3, 8), (4, 5), (4, 6), (4, 7), (5, 4), (5, 5), (5, 6),
                                          (6, 3), (6, 4), (6, 5), (7, 2), (7, 3), (7, 4)])
@pytest.mark.parametrize("log_out", [12, 14])
def test_fold_right_on_the_matrix_cores(hal, oracle, level, log_q, log_out):
    """Rows of 512, 1024 and 2048 bits (vec_len << level) with at least 4096 outputs: fold_right as a GF(2)-linear map on the FP4
    matrix path (csrc/kernels_linmap.hip), every tower level, against the oracle."""
    alloc = hal.dev_alloc()
    assert (1 << log_q) << level in (512, 1024, 2048)
    log_evals = log_out + log_q
    mat = rnd(oracle, 260 + 8 * level + log_q, (1 << log_evals) >> (7 - level))
    vec = rnd(oracle, 261 + level, 1 << log_q)
    out_len = 1 << log_out
    dm, dv, do = upload(hal, alloc, mat), upload(hal, alloc, vec), alloc.alloc(out_len)
    guard = alloc.alloc(8)
    hal.fill(guard, 0x77)
    exp = oracle.arr(out_len)
    hal.fold_right(dm, level, dv, do)
    assert oracle.fold_right(mat, level, vec, exp) == 0
    assert np.array_equal(hal.copy_d2h(do), exp)
    assert (hal.copy_d2h(guard)[:, 0] == 0x77).all()


@pytest.mark.parametrize("log_q", [4, 5, 6])
@pytest.mark.parametrize("log_out", [12, 15])
def test_fold_left_on_the_matrix_cores(hal, oracle, log_q, log_out):
    """fold_left with B32 entries and 16, 32 or 64 vector elements: the gathered rows through csrc/kernels_linmap.hip."""
    alloc = hal.dev_alloc()
    level = 5
    log_evals = log_out + log_q
    mat = rnd(oracle, 360 + log_q, (1 << log_evals) >> (7 - level))
    vec = rnd(oracle, 361 + log_q, 1 << log_q)
    out_len = 1 << log_out
    dm, dv, do = upload(hal, alloc, mat), upload(hal, alloc, vec), alloc.alloc(out_len)
    guard = alloc.alloc(8)
    hal.fill(guard, 0x66)
    exp = oracle.arr(out_len)
    hal.fold_left(dm, level, dv, do)
    assert oracle.fold_left(mat, level, vec, exp) == 0
    assert np.array_equal(hal.copy_d2h(do), exp)
    assert (hal.copy_d2h(guard)[:, 0] == 0x66).all()


def test_fold_validation(hal):
    import binius_amd

    alloc = hal.dev_alloc()
    with pytest.raises(binius_amd.BnError):
        hal.fold_right(alloc.alloc(16), 7, alloc.alloc(2), alloc.alloc(4))  # wrong out len
    with pytest.raises(binius_amd.BnError):
        hal.fold_left(alloc.alloc(16), 9, alloc.alloc(2), alloc.alloc(8))  # bad tower level


# ---- test_generic_compute_composite (layer.rs:773-826)
def test_compute_composite_product(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << 10
    a, b = rnd(oracle, 70, n), rnd(oracle, 71, n)
    da, db, do = upload(hal, alloc, a), upload(hal, alloc, b), alloc.alloc(n)
    expr = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])
    hal.compute_composite([da, db], do, expr)
    assert np.array_equal(hal.copy_d2h(do), oracle.mul_vec(a, b))


def test_compute_composite_generic_circuit(hal, oracle):
    alloc = hal.dev_alloc()
    n = 300
    rows = [rnd(oracle, 72 + i, n) for i in range(3)]
    d = [upload(hal, alloc, r) for r in rows]
    do = alloc.alloc(n)
    c = 0xDEADBEEF_00000000_12345678_9ABCDEF0
    steps = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3), ("const", c), ("mul", 4, 5), ("pow", 6, 5), ("add", 7, 0)]
    expr = hal.compile_expr(steps)
    assert expr.n_vars() == 3
    hal.compute_composite(d, do, expr)
    exp = oracle.arr(n)
    assert oracle.compute_composite(rows, exp, steps) == 0
    assert np.array_equal(hal.copy_d2h(do), exp)


# ---- test_generic_pairwise_product_reduce (layer.rs:909-960)
# (every shape of the subtree walk of kernels_pairtree.hip: a lone launch of 1 .. 6 levels, chains of two and three launches,
# and the hand-over from the element-wise product kernel above 2^15 elements)
@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17])
def test_pairwise_product_reduce(hal, oracle, log_n):
    alloc = hal.dev_alloc()
    n = 1 << log_n
    x = rnd(oracle, 80, n)
    dx = upload(hal, alloc, x)
    outs = [alloc.alloc(n >> (r + 1)) for r in range(log_n)]
    hal.pairwise_product_reduce(dx, outs)
    exp = [oracle.arr(n >> (r + 1)) for r in range(log_n)]
    assert oracle.pairwise_product_reduce(x, exp) == 0
    for o, e in zip(outs, exp):
        assert np.array_equal(hal.copy_d2h(o), e)


def test_pairwise_product_reduce_validation(hal):
    import binius_amd

    alloc = hal.dev_alloc()
    for n, outs in ((1, []), (6, [3]), (8, [4, 2]), (8, [4, 2, 2])):
        with pytest.raises(binius_amd.BnError) as e:
            hal.pairwise_product_reduce(alloc.alloc(n), [alloc.alloc(o) for o in outs])
        assert e.value.kind == "InputValidation"


# ---- test_map_kernels (layer.rs:828-907)
def test_map_kernels_add_assign(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << 10
    a, b = rnd(oracle, 90, n), rnd(oracle, 91, n)
    da, db = upload(hal, alloc, a), upload(hal, alloc, b)

    def kernel(ke, log_chunks, bufs):
        log_len = 10 - log_chunks
        ke.add_assign(log_len, bufs[1].to_ref(), bufs[0])

    hal.map_kernels(kernel, [("chunked_mut", da, 0), ("chunked", db, 0)])
    assert np.array_equal(hal.copy_d2h(da), a ^ b)


# ---- test_generic_kernel_add (layer.rs:412-501)
def test_kernel_add_into_local(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << 10
    a, b = rnd(oracle, 92, n), rnd(oracle, 93, n)
    da, db = upload(hal, alloc, a), upload(hal, alloc, b)
    expr = hal.compile_expr([("var", 0)])

    def kernel(ke, log_chunks, bufs):
        log_len = 10 - log_chunks
        ke.add(log_len, bufs[0].to_ref(), bufs[1].to_ref(), bufs[2])
        acc = ke.decl_value(0)
        ke.sum_composition_evals([bufs[2].to_ref()], expr, 1, acc)
        return [acc]

    (got,) = hal.accumulate_kernels(kernel, [("chunked", da, 3), ("chunked", db, 3), ("local", 10)])
    x = a ^ b
    want = int(np.bitwise_xor.reduce(x[:, 0])) | (int(np.bitwise_xor.reduce(x[:, 1])) << 64)
    assert got == want


# ---- test_generic_single_inner_product_using_kernel_accumulator (layer.rs:329-410)
def test_inner_product_using_kernel_accumulator(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << 8
    a, b = rnd(oracle, 94, n), rnd(oracle, 95, n)
    da, db = upload(hal, alloc, a), upload(hal, alloc, b)
    expr = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])
    coeff = oracle.random_scalars(96, 1)[0]
    init = oracle.random_scalars(97, 1)[0]

    def kernel(ke, log_chunks, bufs):
        acc = ke.decl_value(init)
        ke.sum_composition_evals([bufs[0].to_ref(), bufs[1].to_ref()], expr, coeff, acc)
        return [acc]

    maps = [("chunked", da, 3), ("chunked", db, 3)]
    (got,) = hal.accumulate_kernels(kernel, maps)
    rc, ip = oracle.inner_product(a, 7, b)
    # one logical chunk on this backend: init counted once (CpuLayer with 2^5 chunks would XOR it 32x = 0)
    assert got == init ^ oracle.mul(ip, coeff)
    # oracle run of the same recorded kernel at the same log_chunks
    ops, rets, lc = hal.record(kernel, maps)
    o_ops = [dict(o, steps=o["expr"].steps) if o["op"] == "sum" else o for o in ops]
    rc, want = oracle.run_kernels([("chunked", a, 3), ("chunked", b, 3)], o_ops, rets, lc)
    assert rc == 0 and [got] == want


def test_log_chunks_range():
    """crates/compute/src/layer.rs:827-848 (host only, but needs the library)."""
    import binius_amd

    maps = [("chunked", binius_amd.DevSlice(0x1000, 256), 4), ("chunked_mut", binius_amd.DevSlice(0x9000, 256), 6), ("local", 8)]
    r = binius_amd.log_chunks_range(maps)
    assert (r.start, r.stop) == (0, 2)


# ---- generic circuit in a kernel + materialised Local buffers
def test_accumulate_kernels_generic_circuit(hal, oracle):
    alloc = hal.dev_alloc()
    n = 1 << 9
    a, b, c = rnd(oracle, 100, n), rnd(oracle, 101, n), rnd(oracle, 102, n)
    da, db, dc = (upload(hal, alloc, x) for x in (a, b, c))
    steps = [("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("add", 2, 3)]  # a*b + c
    expr = hal.compile_expr(steps)
    coeff = oracle.random_scalars(103, 1)[0]

    def kernel(ke, log_chunks, bufs):
        ke.add(9 - log_chunks, bufs[0].to_ref(), bufs[1].to_ref(), bufs[3])  # local = a + b
        acc = ke.decl_value(0)
        ke.sum_composition_evals([bufs[3].to_ref(), bufs[1].to_ref(), bufs[2].to_ref()], expr, coeff, acc)
        return [acc]

    maps = [("chunked", da, 0), ("chunked", db, 0), ("chunked", dc, 0), ("local", 9)]
    (got,) = hal.accumulate_kernels(kernel, maps)
    ops, rets, lc = hal.record(kernel, maps)
    o_ops = [dict(o, steps=o["expr"].steps) if o["op"] == "sum" else o for o in ops]
    rc, want = oracle.run_kernels([("chunked", a, 0), ("chunked", b, 0), ("chunked", c, 0), ("local", 9)], o_ops, rets, lc)
    assert rc == 0 and [got] == want


# ---- test_generic_fri_fold (layer.rs:503-570)
@pytest.mark.parametrize("log_batch", [0, 4])
@pytest.mark.parametrize("tw_level", [4, 5])
def test_fri_fold(hal, oracle, log_batch, tw_level):
    import binius_amd

    alloc = hal.dev_alloc()
    log_len, n_fold = 10, 2
    log_domain = log_len
    s_ref = oracle.ntt_s_evals(tw_level, log_domain)
    s_dev = binius_amd.ntt_s_evals(tw_level, log_domain)
    assert np.array_equal(s_ref, s_dev)
    data = rnd(oracle, 110, 1 << (log_len + log_batch))
    challenges = oracle.random_scalars(111, log_batch + n_fold)
    out_len = 1 << (log_len - n_fold)
    din, dout = upload(hal, alloc, data), alloc.alloc(out_len)
    hal.fri_fold(s_dev, tw_level, log_domain, log_len, log_batch, challenges, din, dout)
    got = hal.copy_d2h(dout)
    exp1, exp2 = oracle.arr(out_len), oracle.arr(out_len)
    assert oracle.fri_fold(s_ref, tw_level, log_domain, log_len, log_batch, challenges, data, exp1) == 0
    assert oracle.fold_interleaved(s_ref, tw_level, log_domain, log_len, log_batch, challenges, data, exp2) == 0
    assert np.array_equal(exp1, exp2)
    assert np.array_equal(got, exp1)


@pytest.mark.parametrize("log_len,log_batch,n_fold,tw_level", [(14, 2, 1, 5), (13, 4, 3, 5), (15, 1, 4, 4), (12, 4, 2, 5), (16, 0, 3, 5), (14, 3, 4, 5)])
def test_fri_fold_larger_shapes(hal, oracle, log_len, log_batch, n_fold, tw_level):
    """More arities and sizes (the reference's own test uses log_len 10 with 2 fold rounds)."""
    import binius_amd

    alloc = hal.dev_alloc()
    log_domain = log_len + 1
    s_ref = oracle.ntt_s_evals(tw_level, log_domain)
    s_dev = binius_amd.ntt_s_evals(tw_level, log_domain)
    data = rnd(oracle, 112 + log_len, 1 << (log_len + log_batch))
    challenges = oracle.random_scalars(113 + n_fold, log_batch + n_fold)
    out_len = 1 << (log_len - n_fold)
    din, dout = upload(hal, alloc, data), alloc.alloc(out_len)
    hal.fri_fold(s_dev, tw_level, log_domain, log_len, log_batch, challenges, din, dout)
    exp = oracle.arr(out_len)
    assert oracle.fri_fold(s_ref, tw_level, log_domain, log_len, log_batch, challenges, data, exp) == 0
    assert np.array_equal(hal.copy_d2h(dout), exp)
    assert np.array_equal(hal.copy_d2h(din), data)  # the input is read-only


def test_fri_fold_validation(hal, oracle):
    import binius_amd

    alloc = hal.dev_alloc()
    s = binius_amd.ntt_s_evals(5, 10)
    with pytest.raises(binius_amd.BnError):
        hal.fri_fold(s, 5, 10, 10, 0, [1, 2], alloc.alloc(1 << 9), alloc.alloc(1 << 8))  # bad in len
    with pytest.raises(binius_amd.BnError):
        hal.fri_fold(s, 5, 10, 4, 2, [1], alloc.alloc(1 << 6), alloc.alloc(1 << 4))  # too few challenges
    with pytest.raises(binius_amd.BnError):
        hal.fri_fold(s, 5, 10, 4, 0, [1, 2], alloc.alloc(1 << 4), alloc.alloc(1 << 3))  # bad out len


def test_context_is_safe_to_call_from_several_threads(hal, oracle):
    """The trait lets the host call a layer from several threads (rayon join/map); entry points of
    one context are serialised by the library."""
    import threading

    alloc = hal.dev_alloc()
    n = 1 << 12
    jobs = []
    for t in range(8):
        e0, e1 = rnd(oracle, 900 + t, n), rnd(oracle, 950 + t, n)
        z = oracle.random_scalars(970 + t, 1)[0]
        jobs.append((upload(hal, alloc, e0), upload(hal, alloc, e1), e0, e1, z))
    errs = []

    def work(job):
        d0, d1, e0, e1, z = job
        try:
            for _ in range(5):
                hal.extrapolate_line(d0, d1, z)
                exp = e0
                oracle.extrapolate_line(exp, e1, z)
                if not np.array_equal(hal.copy_d2h(d0), exp):
                    errs.append("mismatch")
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs


def test_fresh_thread_allocating_entry_points(oracle):
    """per-call allocations (NTT scratch and tables, fri_fold scratch, pinned gather staging,
    Local kernel buffers) happen on the CALLING thread's current device; every entry point therefore makes
    the context's device current first.  A freshly spawned thread that has never touched HIP drives a
    fresh context through the lazily allocating paths; with two GPUs the context lives on the last one."""
    import threading

    import torch

    import binius_amd

    dev = torch.cuda.device_count() - 1
    out = {}

    def work():
        try:
            ctx = binius_amd.Context(dev, 1 << 18)
            alloc = ctx.dev_alloc()
            # NTT (scratch + cached tables)
            log_y = 14
            s = binius_amd.ntt_s_evals(5, log_y)
            data = oracle.splitmix_words(0x77, (1 << log_y) // 2).view(np.uint32).copy()
            want = data.copy()
            assert oracle.ntt_forward(want, 5, 5, oracle.ntt_s_evals(5, log_y), log_y, 0, log_y, 0) == 0
            d = alloc.alloc((1 << log_y) // 4)
            ctx.copy_h2d(data.view(np.uint64).reshape(-1, 2), d)
            ctx.ntt_forward(d.ptr, 5, 5, s, log_y, 0, log_y, 0)
            got = ctx.copy_d2h(d).reshape(-1).view(np.uint32)
            out["ntt"] = bool(np.array_equal(got, want))
            # inner product over a subfield (scratch) and a fold
            a, b = oracle.random_b128(1, 1 << 6), oracle.random_b128(2, 1 << 10)
            da, db = alloc.alloc(1 << 6), alloc.alloc(1 << 10)
            ctx.copy_h2d(a, da)
            ctx.copy_h2d(b, db)
            out["ip"] = ctx.inner_product(da, 3, db) == oracle.inner_product(a, 3, b)[1]
            ctx.close()
        except Exception as ex:  # noqa: BLE001
            out["err"] = repr(ex)

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert out == {"ntt": True, "ip": True}, out




def test_device_numa_node_and_thread_binding(hal):
    """bn_device_numa_node reads the device's PCI function from sysfs; the helper only ever narrows the affinity."""
    import os

    import binius_amd

    node = binius_amd.device_numa_node(0)
    assert node is None or (isinstance(node, int) and node >= 0)
    before = os.sched_getaffinity(0)
    try:
        what = binius_amd.bind_host_thread_to_device(0)
        after = os.sched_getaffinity(0)
        assert after <= before and len(after) >= 1
        if node is not None and what.startswith("NUMA node"):
            cpus = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
            assert what == "NUMA node %d (%d CPUs)" % (node, len(after)) and cpus
    finally:
        os.sched_setaffinity(0, before)
