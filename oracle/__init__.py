"""ctypes bindings for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's algorithms for the hot path (see the
headers of oracle/*.h for the reference file:line each function follows).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package, and
only as the checker.  Nothing under ``binius_amd/`` imports it.

Field elements: a BinaryField128b array is a numpy ``uint64`` array of shape (n, 2) = (lo, hi),
which is exactly the little-endian u128 memory layout of the reference
(crates/field/src/binary_field.rs:747, 118).  Scalars are Python ints.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile the C restatement (gcc) if needed."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class B128(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


class Step(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint64), ("cst", B128)]


class MemMap(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("log_min_chunk_size", C.c_uint32),
        ("data", C.c_void_p),
        ("len", C.c_uint64),
        ("log_size", C.c_uint32),
    ]


class KSlice(C.Structure):
    _fields_ = [("buf", C.c_uint32), ("off", C.c_uint64), ("len", C.c_uint64)]


class KOp(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("value", C.c_uint32),
        ("scalar", B128),
        ("steps", C.POINTER(Step)),
        ("n_steps", C.c_uint32),
        ("n_rows", C.c_uint32),
        ("rows", C.POINTER(KSlice)),
        ("src1", KSlice),
        ("src2", KSlice),
        ("dst", KSlice),
    ]


STEP_ADD, STEP_MUL, STEP_POW, STEP_CONST, STEP_VAR = range(5)
MAP_CHUNKED, MAP_CHUNKED_MUT, MAP_LOCAL = range(3)
KOP_DECL_VALUE, KOP_SUM_COMPOSITION, KOP_ADD, KOP_ADD_ASSIGN = range(4)

MASK64 = (1 << 64) - 1


def to_b128(x):
    x = int(x)
    return B128(x & MASK64, (x >> 64) & MASK64)


def from_b128(b):
    return int(b.lo) | (int(b.hi) << 64)


def arr(n):
    return np.zeros((n, 2), dtype=np.uint64)


def ints_to_arr(vals):
    a = arr(len(vals))
    for i, v in enumerate(vals):
        a[i, 0] = v & MASK64
        a[i, 1] = (v >> 64) & MASK64
    return a


def arr_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 2)
    return [int(a[i, 0]) | (int(a[i, 1]) << 64) for i in range(a.shape[0])]


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(B128))


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ref_gf_mul.restype = C.c_uint64
        L.ref_gf_mul.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.ref_gf_mul_slow.restype = C.c_uint64
        L.ref_gf_mul_slow.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.ref_gf_mul_alpha.restype = C.c_uint64
        L.ref_gf_mul_alpha.argtypes = [C.c_uint64, C.c_int]
        L.ref_gf_square.restype = C.c_uint64
        L.ref_gf_square.argtypes = [C.c_uint64, C.c_int]
        L.ref_gf_invert.restype = C.c_uint64
        L.ref_gf_invert.argtypes = [C.c_uint64, C.c_int]
        for name in ("ref_b128_mul_p",):
            getattr(L, name).argtypes = [C.POINTER(B128)] * 3
        L.ref_b128_square_p.argtypes = [C.POINTER(B128)] * 2
        L.ref_b128_invert_p.argtypes = [C.POINTER(B128)] * 2
        L.ref_b128_mul_subfield_p.argtypes = [C.POINTER(B128), C.POINTER(B128), C.c_int, C.POINTER(B128)]
        L.ref_b128_mul_vec.argtypes = [C.POINTER(B128)] * 3 + [C.c_size_t]
        L.ref_splitmix_fill.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), C.c_size_t]
        L.ref_circuit_eval.restype = B128
        L.ref_circuit_eval.argtypes = [C.POINTER(Step), C.c_size_t, C.POINTER(B128)]
        L.ref_extrapolate_line.argtypes = [C.POINTER(B128), C.POINTER(B128), C.c_size_t, C.c_size_t, B128]
        L.ref_tensor_expand.argtypes = [C.POINTER(B128), C.c_size_t, C.c_size_t, C.POINTER(B128), C.c_size_t, C.c_int]
        L.ref_inner_product.argtypes = [C.POINTER(B128), C.c_size_t, C.c_int, C.POINTER(B128), C.c_size_t, C.POINTER(B128)]
        for name in ("ref_fold_left", "ref_fold_right"):
            getattr(L, name).argtypes = [
                C.POINTER(B128), C.c_size_t, C.c_int, C.POINTER(B128), C.c_size_t, C.POINTER(B128), C.c_size_t,
            ]
        L.ref_compute_composite.argtypes = [
            C.POINTER(C.POINTER(B128)), C.c_size_t, C.c_size_t, C.POINTER(B128), C.c_size_t,
            C.POINTER(Step), C.c_size_t, C.c_size_t,
        ]
        L.ref_pairwise_product_reduce.argtypes = [
            C.POINTER(B128), C.c_size_t, C.POINTER(C.POINTER(B128)), C.POINTER(C.c_size_t), C.c_size_t,
        ]
        L.ref_add_assign.argtypes = [C.POINTER(B128), C.POINTER(B128), C.c_size_t]
        L.ref_log_chunks_range.argtypes = [C.POINTER(MemMap), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ref_run_kernels.argtypes = [
            C.POINTER(MemMap), C.c_size_t, C.POINTER(KOp), C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t,
            C.POINTER(B128), C.c_int,
        ]
        L.ref_ntt_s_evals.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.ref_ntt_twiddle.restype = C.c_uint64
        L.ref_ntt_twiddle.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_uint64]
        L.ref_ntt_get_subspace_eval.restype = C.c_uint64
        L.ref_ntt_get_subspace_eval.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_uint64]
        for name in ("ref_ntt_forward", "ref_ntt_inverse"):
            getattr(L, name).argtypes = [
                C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int, C.c_int,
                C.c_uint64, C.c_int, C.c_int,
            ]
        for name in ("ref_fri_fold", "ref_fold_interleaved"):
            getattr(L, name).argtypes = [
                C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(B128), C.c_size_t,
                C.POINTER(B128), C.c_size_t, C.POINTER(B128), C.c_size_t,
            ]
        L.ref_round_evals.argtypes = [
            C.POINTER(C.POINTER(B128)), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.c_size_t, B128,
            C.POINTER(B128), C.c_int,
        ]
        L.ref_fold_high.argtypes = [C.POINTER(B128), C.c_uint, B128, C.c_int]
        L.ref_bivariate_sumcheck_prove.argtypes = [
            C.POINTER(C.POINTER(B128)), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.c_size_t,
            C.POINTER(B128), B128, C.POINTER(B128), C.POINTER(B128), C.POINTER(B128), C.c_int,
        ]
        L.ref_round_evals_eq.argtypes = [
            C.POINTER(C.POINTER(B128)), C.c_size_t, C.c_uint, C.POINTER(B128), C.POINTER(C.c_uint32), C.c_size_t, B128,
            C.POINTER(B128),
        ]
        L.ref_bivariate_mlecheck_prove.argtypes = [
            C.POINTER(C.POINTER(B128)), C.c_size_t, C.c_uint, C.POINTER(B128), C.POINTER(B128), C.POINTER(C.c_uint32),
            C.c_size_t, C.POINTER(B128), B128, C.POINTER(B128), C.POINTER(B128), C.POINTER(B128),
        ]
        L.ref_mle_evaluate.restype = B128
        L.ref_mle_evaluate.argtypes = [C.POINTER(B128), C.c_uint, C.POINTER(B128)]
        U8P = C.POINTER(C.c_uint8)
        L.ref_groestl256.argtypes = [C.c_char_p, C.c_size_t, U8P]
        L.ref_groestl256_compress2.argtypes = [C.c_char_p, C.c_char_p, U8P]
        L.ref_merkle_build.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.ref_merkle_root_from_branch.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint32, U8P]
        L.ref_fast_bivariate_sumcheck_prove.argtypes = [
            C.POINTER(C.POINTER(B128)), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(B128), B128,
            C.POINTER(B128), C.POINTER(B128), C.POINTER(B128), C.POINTER(B128), C.POINTER(B128), C.c_int,
        ]
        L.ref_fast_inner_product.argtypes = [C.POINTER(B128), C.POINTER(B128), C.c_size_t, C.POINTER(B128), C.POINTER(B128), C.POINTER(B128), C.c_int]
        L.ref_evaluate_univariate.restype = B128
        L.ref_evaluate_univariate.argtypes = [C.POINTER(B128), C.c_size_t, B128]
        _lib = L
    return _lib


# ------------------------------------------------------------------ field
def gf_mul(a, b, level):
    return int(lib().ref_gf_mul(a, b, level))


def gf_mul_slow(a, b, level):
    return int(lib().ref_gf_mul_slow(a, b, level))


def gf_square(a, level):
    return int(lib().ref_gf_square(a, level))


def gf_invert(a, level):
    return int(lib().ref_gf_invert(a, level))


def gf_mul_alpha(a, level):
    return int(lib().ref_gf_mul_alpha(a, level))


def mul(a, b):
    """BinaryField128b product of two Python ints."""
    x, y, o = to_b128(a), to_b128(b), B128()
    lib().ref_b128_mul_p(C.byref(x), C.byref(y), C.byref(o))
    return from_b128(o)


def square(a):
    x, o = to_b128(a), B128()
    lib().ref_b128_square_p(C.byref(x), C.byref(o))
    return from_b128(o)


def invert(a):
    x, o = to_b128(a), B128()
    lib().ref_b128_invert_p(C.byref(x), C.byref(o))
    return from_b128(o)


def mul_subfield(a, s, iota):
    x, y, o = to_b128(a), to_b128(s), B128()
    lib().ref_b128_mul_subfield_p(C.byref(x), C.byref(y), iota, C.byref(o))
    return from_b128(o)


def mul_vec(a, b):
    out = arr(a.shape[0])
    lib().ref_b128_mul_vec(_p(a), _p(b), _p(out), a.shape[0])
    return out


def splitmix_words(seed, n_words):
    out = np.zeros(n_words, dtype=np.uint64)
    lib().ref_splitmix_fill(seed, out.ctypes.data_as(C.POINTER(C.c_uint64)), n_words)
    return out


def random_b128(seed, n):
    """n uniform BinaryField128b elements: SplitMix64(seed), two draws per element (lo, hi)."""
    return splitmix_words(seed, 2 * n).reshape(n, 2)


def random_scalars(seed, n):
    return arr_to_ints(random_b128(seed, n))


# ------------------------------------------------------------------ circuits
def make_steps(steps):
    """steps: list of ('add',l,r) | ('mul',l,r) | ('pow',base,exp) | ('const',value) | ('var',idx)."""
    kinds = {"add": STEP_ADD, "mul": STEP_MUL, "pow": STEP_POW, "const": STEP_CONST, "var": STEP_VAR}
    out = (Step * max(1, len(steps)))()
    for i, s in enumerate(steps):
        k = kinds[s[0]]
        out[i].kind = k
        if k in (STEP_ADD, STEP_MUL, STEP_POW):
            out[i].a, out[i].b = s[1], s[2]
        elif k == STEP_CONST:
            out[i].cst = to_b128(s[1])
        else:
            out[i].a = s[1]
    return out


def circuit_eval(steps, query):
    st = make_steps(steps)
    q = ints_to_arr(list(query))
    return from_b128(lib().ref_circuit_eval(st, len(steps), _p(q)))


# ------------------------------------------------------------------ layer ops
def extrapolate_line(evals_0, evals_1, z):
    rc = lib().ref_extrapolate_line(_p(evals_0), _p(evals_1), evals_0.shape[0], evals_1.shape[0], to_b128(z))
    return rc


def tensor_expand(data, log_n, coords, assign=True):
    c = ints_to_arr(list(coords))
    return lib().ref_tensor_expand(_p(data), data.shape[0], log_n, _p(c), len(coords), 1 if assign else 0)


def inner_product(a, tower_level, b):
    o = B128()
    rc = lib().ref_inner_product(_p(a), a.shape[0], tower_level, _p(b), b.shape[0], C.byref(o))
    return rc, from_b128(o)


def fold_left(mat, tower_level, vec, out):
    return lib().ref_fold_left(_p(mat), mat.shape[0], tower_level, _p(vec), vec.shape[0], _p(out), out.shape[0])


def fold_right(mat, tower_level, vec, out):
    return lib().ref_fold_right(_p(mat), mat.shape[0], tower_level, _p(vec), vec.shape[0], _p(out), out.shape[0])


def compute_composite(inputs, out, steps, n_vars=None):
    n_rows = len(inputs)
    ptrs = (C.POINTER(B128) * max(1, n_rows))(*[_p(a) for a in inputs])
    st = make_steps(steps)
    row_len = inputs[0].shape[0] if n_rows else 0
    if n_vars is None:
        n_vars = n_rows
    return lib().ref_compute_composite(ptrs, n_rows, row_len, _p(out), out.shape[0], st, len(steps), n_vars)


def pairwise_product_reduce(inp, round_outputs):
    n_r = len(round_outputs)
    ptrs = (C.POINTER(B128) * max(1, n_r))(*[_p(a) for a in round_outputs])
    lens = (C.c_size_t * max(1, n_r))(*[a.shape[0] for a in round_outputs])
    return lib().ref_pairwise_product_reduce(_p(inp), inp.shape[0], ptrs, lens, n_r)


def log_chunks_range(maps):
    mm = _make_maps(maps)
    s, e = C.c_uint32(), C.c_uint32()
    rc = lib().ref_log_chunks_range(mm, len(maps), C.byref(s), C.byref(e))
    return rc, s.value, e.value


def _make_maps(maps):
    """maps: list of ('chunked', arr, log_min) | ('chunked_mut', arr, log_min) | ('local', log_size)."""
    mm = (MemMap * max(1, len(maps)))()
    for i, m in enumerate(maps):
        if m[0] == "local":
            mm[i].kind = MAP_LOCAL
            mm[i].log_size = m[1]
        else:
            mm[i].kind = MAP_CHUNKED if m[0] == "chunked" else MAP_CHUNKED_MUT
            mm[i].data = m[1].ctypes.data
            mm[i].len = m[1].shape[0]
            mm[i].log_min_chunk_size = m[2]
    return mm


def run_kernels(maps, ops, ret_values, log_chunks):
    """ops: list of dicts: {'op':'decl','value':id,'init':int} |
    {'op':'sum','value':id,'steps':[...],'coeff':int,'rows':[(buf,off,len),...]} |
    {'op':'add','src1':(b,o,l),'src2':(b,o,l),'dst':(b,o,l)} | {'op':'add_assign','src':..., 'dst':...}"""
    mm = _make_maps(maps)
    kops = (KOp * max(1, len(ops)))()
    keep = []
    for i, o in enumerate(ops):
        if o["op"] == "decl":
            kops[i].kind = KOP_DECL_VALUE
            kops[i].value = o["value"]
            kops[i].scalar = to_b128(o["init"])
        elif o["op"] == "sum":
            kops[i].kind = KOP_SUM_COMPOSITION
            kops[i].value = o["value"]
            kops[i].scalar = to_b128(o["coeff"])
            st = make_steps(o["steps"])
            rows = (KSlice * max(1, len(o["rows"])))(*[KSlice(*r) for r in o["rows"]])
            keep += [st, rows]
            kops[i].steps = st
            kops[i].n_steps = len(o["steps"])
            kops[i].rows = rows
            kops[i].n_rows = len(o["rows"])
        elif o["op"] == "add":
            kops[i].kind = KOP_ADD
            kops[i].src1, kops[i].src2, kops[i].dst = KSlice(*o["src1"]), KSlice(*o["src2"]), KSlice(*o["dst"])
        elif o["op"] == "add_assign":
            kops[i].kind = KOP_ADD_ASSIGN
            kops[i].src1, kops[i].dst = KSlice(*o["src"]), KSlice(*o["dst"])
        else:
            raise ValueError(o)
    rv = (C.c_uint32 * max(1, len(ret_values)))(*ret_values)
    out = arr(max(1, len(ret_values)))
    rc = lib().ref_run_kernels(mm, len(maps), kops, len(ops), rv, len(ret_values), _p(out), log_chunks)
    return rc, arr_to_ints(out)[: len(ret_values)]


# ------------------------------------------------------------------ NTT
NTT_MAX_DIM = 64


def ntt_s_evals(level, log_domain):
    s = np.zeros(NTT_MAX_DIM * NTT_MAX_DIM, dtype=np.uint64)
    rc = lib().ref_ntt_s_evals(level, log_domain, s.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == 0, "bad NTT domain"
    return s


def _sp(s):
    return s.ctypes.data_as(C.POINTER(C.c_uint64))


def ntt_twiddle(s_evals, log_domain, layer, index):
    return int(lib().ref_ntt_twiddle(_sp(s_evals), log_domain, layer, index))


def ntt_get_subspace_eval(s_evals, log_domain, i, j):
    return int(lib().ref_ntt_get_subspace_eval(_sp(s_evals), log_domain, i, j))


def ntt_forward(data, elem_level, tw_level, s_evals, log_domain, log_x, log_y, log_z, coset=0, coset_bits=0, skip_rounds=0):
    return lib().ref_ntt_forward(
        data.ctypes.data, elem_level, tw_level, _sp(s_evals), log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds
    )


def ntt_inverse(data, elem_level, tw_level, s_evals, log_domain, log_x, log_y, log_z, coset=0, coset_bits=0, skip_rounds=0):
    return lib().ref_ntt_inverse(
        data.ctypes.data, elem_level, tw_level, _sp(s_evals), log_domain, log_x, log_y, log_z, coset, coset_bits, skip_rounds
    )


def fri_fold(s_evals, tw_level, log_domain, log_len, log_batch, challenges, data_in, data_out):
    ch = ints_to_arr(list(challenges))
    return lib().ref_fri_fold(
        _sp(s_evals), tw_level, log_domain, log_len, log_batch, _p(ch), len(challenges),
        _p(data_in), data_in.shape[0], _p(data_out), data_out.shape[0],
    )


def fold_interleaved(s_evals, tw_level, log_domain, log_len, log_batch, challenges, codeword, out):
    ch = ints_to_arr(list(challenges))
    return lib().ref_fold_interleaved(
        _sp(s_evals), tw_level, log_domain, log_len, log_batch, _p(ch), len(challenges),
        _p(codeword), codeword.shape[0], _p(out), out.shape[0],
    )


# ------------------------------------------------------------------ sumcheck
def round_evals(multilins, n_vars, comps, batch_coeff, threads=1):
    ptrs = (C.POINTER(B128) * len(multilins))(*[_p(a) for a in multilins])
    flat = [i for pair in comps for i in pair]
    cc = (C.c_uint32 * max(1, len(flat)))(*flat)
    out = arr(2)
    rc = lib().ref_round_evals(ptrs, len(multilins), n_vars, cc, len(comps), to_b128(batch_coeff), _p(out), threads)
    return rc, arr_to_ints(out)


def fold_high(evals, n_vars, z, threads=1):
    return lib().ref_fold_high(_p(evals), n_vars, to_b128(z), threads)


def bivariate_sumcheck_prove(multilins, n_vars, comps, sums, batch_coeff, challenges, threads=1):
    """multilins (list of arrays) are folded in place. Returns (round_coeffs[n_vars][3], final_evals[m])."""
    ptrs = (C.POINTER(B128) * len(multilins))(*[_p(a) for a in multilins])
    flat = [i for pair in comps for i in pair]
    cc = (C.c_uint32 * max(1, len(flat)))(*flat)
    s = ints_to_arr(list(sums)) if len(sums) else arr(1)
    ch = ints_to_arr(list(challenges))
    rc_out = arr(3 * n_vars)
    fe = arr(len(multilins))
    rc = lib().ref_bivariate_sumcheck_prove(
        ptrs, len(multilins), n_vars, cc, len(comps), _p(s), to_b128(batch_coeff), _p(ch), _p(rc_out), _p(fe), threads
    )
    assert rc == 0
    co = arr_to_ints(rc_out)
    return [co[3 * r : 3 * r + 3] for r in range(n_vars)], arr_to_ints(fe)


def round_evals_eq(multilins, n_vars, eq_ind, comps, batch_coeff):
    """MLE-check round evaluations (v3/bivariate_mlecheck.rs:391-520): compositions a*b*eq_ind."""
    ptrs = (C.POINTER(B128) * len(multilins))(*[_p(a) for a in multilins])
    flat = [i for pair in comps for i in pair]
    cc = (C.c_uint32 * max(1, len(flat)))(*flat)
    out = arr(2)
    rc = lib().ref_round_evals_eq(ptrs, len(multilins), n_vars, _p(eq_ind), cc, len(comps), to_b128(batch_coeff), _p(out))
    return rc, arr_to_ints(out)


def bivariate_mlecheck_prove(multilins, n_vars, eq_ind, eq_ind_challenges, comps, sums, batch_coeff, challenges):
    """multilins and eq_ind (arrays) are folded in place.  Returns (round_coeffs[n_vars][4],
    final_evals[m + 1]) -- the last final value is eq_ind_prefix_eval."""
    ptrs = (C.POINTER(B128) * len(multilins))(*[_p(a) for a in multilins])
    flat = [i for pair in comps for i in pair]
    cc = (C.c_uint32 * max(1, len(flat)))(*flat)
    s = ints_to_arr(list(sums)) if len(sums) else arr(1)
    ch = ints_to_arr(list(challenges))
    eqc = ints_to_arr(list(eq_ind_challenges))
    rc_out = arr(4 * n_vars)
    fe = arr(len(multilins) + 1)
    rc = lib().ref_bivariate_mlecheck_prove(
        ptrs, len(multilins), n_vars, _p(eq_ind), _p(eqc), cc, len(comps), _p(s), to_b128(batch_coeff), _p(ch), _p(rc_out), _p(fe)
    )
    assert rc == 0
    co = arr_to_ints(rc_out)
    return [co[4 * r : 4 * r + 4] for r in range(n_vars)], arr_to_ints(fe)


def mle_evaluate(evals, n_vars, point):
    p = ints_to_arr(list(point)) if len(point) else arr(1)
    return from_b128(lib().ref_mle_evaluate(_p(evals), n_vars, _p(p)))


def evaluate_univariate(coeffs, x):
    c = ints_to_arr(list(coeffs))
    return from_b128(lib().ref_evaluate_univariate(_p(c), len(coeffs), to_b128(x)))


# ---- Groestl-256 and the binary Merkle tree (oracle/merkle_ref.h)
def groestl256(msg):
    out = (C.c_uint8 * 32)()
    lib().ref_groestl256(bytes(msg), len(msg), out)
    return bytes(out)


def groestl256_compress2(left, right):
    assert len(left) == 32 and len(right) == 32
    out = (C.c_uint8 * 32)()
    lib().ref_groestl256_compress2(bytes(left), bytes(right), out)
    return bytes(out)


def merkle_build(elems, batch_size):
    """elems: (n, 2) uint64 array of BinaryField128b.  Returns (rc, nodes) with nodes a
    (2 * n_leaves - 1, 32) uint8 array: layers flattened leaves first, root last."""
    elems = np.ascontiguousarray(elems, dtype=np.uint64)
    n = elems.shape[0]
    n_leaves = n // batch_size if batch_size else 0
    nodes = np.zeros((max(2 * n_leaves - 1, 1), 32), dtype=np.uint8)
    rc = lib().ref_merkle_build(elems.ctypes.data, n, batch_size, nodes.ctypes.data)
    return rc, nodes


def merkle_root_from_branch(leaf_digest, index, branch):
    out = (C.c_uint8 * 32)()
    lib().ref_merkle_root_from_branch(bytes(leaf_digest), index, b"".join(bytes(b) for b in branch), len(branch), out)
    return bytes(out)


# ---- optimized CPU version of the sumcheck loop (oracle/fastcpu_ref.c): the cpu_baseline of bench.py
_POLYVAL = None


def _polyval_tables():
    """The reference's BINARY_TO_POLYVAL_TRANSFORMATION and its inverse (crates/field/src/polyval.rs:516-784),
    from the committed golden data."""
    global _POLYVAL
    if _POLYVAL is None:
        import json

        kats = json.load(open(os.path.join(os.path.dirname(_HERE), "tests", "golden", "field_kats.json")))
        _POLYVAL = (ints_to_arr([int(x, 16) for x in kats["binary_to_polyval"]]), ints_to_arr([int(x, 16) for x in kats["polyval_to_binary"]]))
    return _POLYVAL


def fast_bivariate_sumcheck_prove(multilins, n_vars, comps, sums, batch_coeff, challenges, threads=1):
    """Same transcript as bivariate_sumcheck_prove, computed in the isomorphic POLYVAL field with PCLMULQDQ
    and OpenMP.  multilins are overwritten.  Returns None when the host has no PCLMULQDQ."""
    fwd, inv = _polyval_tables()
    ptrs = (C.POINTER(B128) * len(multilins))(*[_p(a) for a in multilins])
    flat = [i for pair in comps for i in pair]
    cc = (C.c_uint32 * max(1, len(flat)))(*flat)
    s = ints_to_arr(list(sums)) if len(sums) else arr(1)
    ch = ints_to_arr(list(challenges))
    rc_out = arr(3 * n_vars)
    fe = arr(len(multilins))
    rc = lib().ref_fast_bivariate_sumcheck_prove(
        ptrs, len(multilins), n_vars, cc, len(comps), _p(s), to_b128(batch_coeff), _p(ch), _p(fwd), _p(inv), _p(rc_out), _p(fe), threads
    )
    if rc == 2:
        return None
    assert rc == 0
    co = arr_to_ints(rc_out)
    return [co[3 * r : 3 * r + 3] for r in range(n_vars)], arr_to_ints(fe)


def fast_inner_product(a, b, threads=1):
    """XOR_i a[i] * b[i] over GF(2^128), POLYVAL-basis PCLMULQDQ arithmetic + OpenMP (inputs untouched).  None when the
    host has no PCLMULQDQ."""
    fwd, inv = _polyval_tables()
    assert a.shape == b.shape
    out = arr(1)
    rc = lib().ref_fast_inner_product(_p(a), _p(b), a.shape[0], _p(fwd), _p(inv), _p(out), threads)
    if rc == 2:
        return None
    assert rc == 0
    return arr_to_ints(out)[0]


# ------------------------------------------------------------------ old HAL (crates/hal) restatement: hal_ref.c
ORDER_LOW_TO_HIGH, ORDER_HIGH_TO_LOW = 0, 1
HAL_ML_FOLDED, HAL_ML_TRANSPARENT = 0, 1


class HalMultilinear(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("tower_level", C.c_uint32),
        ("evals", C.c_void_p),
        ("len", C.c_uint64),
        ("suffix_eval", B128),
        ("n_vars_ml", C.c_uint32),
    ]


class HalEvaluator(C.Structure):
    _fields_ = [
        ("composition", C.POINTER(Step)),
        ("n_steps", C.c_uint64),
        ("composition_at_infinity", C.POINTER(Step)),
        ("n_steps_inf", C.c_uint64),
        ("eval_point_start", C.c_uint32),
        ("eval_point_end", C.c_uint32),
        ("eq_ind", C.c_void_p),
    ]


def _hal_ml(ml):
    """ml: ('folded', evals_array, suffix_eval) | ('transparent', packed_array, tower_level, n_vars_ml)."""
    m = HalMultilinear()
    if ml[0] == "folded":
        m.kind, m.evals, m.len, m.suffix_eval = HAL_ML_FOLDED, ml[1].ctypes.data, ml[1].shape[0], to_b128(ml[2])
    else:
        m.kind, m.evals, m.len, m.tower_level, m.n_vars_ml = HAL_ML_TRANSPARENT, ml[1].ctypes.data, ml[1].shape[0], ml[2], ml[3]
    return m


def hal_round_evals(order, n_vars, tensor_query, multilinears, evaluators, nontrivial_points):
    """evaluators: list of dicts {steps, steps_inf, start, end, eq_ind (array or None)}.  Returns (rc, [[values per point] per evaluator])."""
    L = lib()
    L.ref_hal_round_evals.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(HalMultilinear), C.c_uint32, C.POINTER(HalEvaluator),
                                      C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    mls = (HalMultilinear * max(1, len(multilinears)))(*[_hal_ml(m) for m in multilinears])
    keep = []
    evs = (HalEvaluator * max(1, len(evaluators)))()
    total = 0
    for i, e in enumerate(evaluators):
        st, si = make_steps(e["steps"]), make_steps(e["steps_inf"])
        keep += [st, si]
        evs[i].composition, evs[i].n_steps = st, len(e["steps"])
        evs[i].composition_at_infinity, evs[i].n_steps_inf = si, len(e["steps_inf"])
        evs[i].eval_point_start, evs[i].eval_point_end = e["start"], e["end"]
        evs[i].eq_ind = e["eq_ind"].ctypes.data if e.get("eq_ind") is not None else None
        total += e["end"] - e["start"]
    q = tensor_query if tensor_query is not None else arr(1)
    query_vars = (q.shape[0].bit_length() - 1) if tensor_query is not None else 0
    pts = ints_to_arr(list(nontrivial_points)) if len(nontrivial_points) else arr(1)
    out = arr(max(1, total))
    rc = L.ref_hal_round_evals(order, n_vars, q.ctypes.data, query_vars, mls, len(multilinears), evs, len(evaluators), pts.ctypes.data,
                               len(nontrivial_points), out.ctypes.data)
    vals = arr_to_ints(out)
    res, off = [], 0
    for e in evaluators:
        res.append(vals[off : off + e["end"] - e["start"]])
        off += e["end"] - e["start"]
    return rc, res


def hal_fold_multilinear(order, n_vars, multilinear, challenge, tensor_query=None):
    """Returns (rc, folded evaluations as an (n, 2) uint64 array)."""
    L = lib()
    L.ref_hal_fold_multilinear.argtypes = [C.c_int, C.c_uint32, C.POINTER(HalMultilinear), B128, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                           C.POINTER(C.c_uint64)]
    m = _hal_ml(multilinear)
    q = tensor_query if tensor_query is not None else arr(1)
    query_vars = (q.shape[0].bit_length() - 1) if tensor_query is not None else 0
    out = arr(1 << n_vars)
    n = C.c_uint64(0)
    rc = L.ref_hal_fold_multilinear(order, n_vars, C.byref(m), to_b128(challenge), q.ctypes.data, query_vars, out.ctypes.data, out.shape[0], C.byref(n))
    return rc, out[: n.value].copy()
