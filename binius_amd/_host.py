"""ctypes binding of libbinius_amd_host.so -- the compiled C++ host mirror
(binius_amd/host/{compute_layer,sumcheck}.hpp): a whole BivariateSumcheckProver run behind one C
call, i.e. the prover loop driven at compiled-host speed over the same C ABI."""
import ctypes as C
import os

from ._ffi import F128, BnError, _f128_array, from_f128, lib, to_f128

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbinius_amd_host.so")
_lib = None

REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(F128))


def host_lib():
    global _lib
    if _lib is None:
        lib()  # libbinius_amd.so first (dependency, resolved through rpath as well)
        if not os.path.exists(_SO):
            raise ImportError("binius_amd: %s is missing -- run __graft_entry__.build()" % _SO)
        L = C.CDLL(_SO)
        L.bnh_last_error.restype = C.c_char_p
        L.bnh_bivariate_sumcheck_prove.restype = C.c_int
        L.bnh_bivariate_sumcheck_prove.argtypes = [
            C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p, C.c_uint64, C.c_uint32,
            C.POINTER(C.c_uint32), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128),
            REDUCE_FN, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
        ]
        L.bnh_shm_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.bnh_shm_close.argtypes = [C.c_void_p]
        L.bnh_shm_allgather.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64)]
        L.bnh_bivariate_mlecheck_prove.restype = C.c_int
        L.bnh_bivariate_mlecheck_prove.argtypes = [
            C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(F128), C.c_void_p, C.c_uint64,
            C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128),
        ]
        L.bnh_mlecheck_new.restype = C.c_int
        L.bnh_mlecheck_new.argtypes = [
            C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(F128), C.c_void_p, C.c_uint64,
            C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(F128), C.POINTER(C.c_void_p),
        ]
        L.bnh_mlecheck_execute.argtypes = [C.c_void_p, C.POINTER(F128), C.POINTER(F128)]
        L.bnh_mlecheck_fold.argtypes = [C.c_void_p, C.POINTER(F128)]
        L.bnh_mlecheck_finish.argtypes = [C.c_void_p, C.POINTER(F128)]
        L.bnh_mlecheck_free.argtypes = [C.c_void_p]
        L.bnh_mlecheck_free.restype = None
        L.bnh_mlecheck_last_mode.restype = C.c_int
        L.bnh_fri_commit_fold.restype = C.c_int
        L.bnh_fri_commit_fold.argtypes = [
            C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
            C.c_uint64, C.POINTER(F128), C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
        ]
        L.bnh_batch_sumcheck_prove.restype = C.c_int
        L.bnh_batch_sumcheck_prove.argtypes = [
            C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(F128), C.c_void_p, C.c_uint64,
            C.POINTER(F128), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128),
        ]
        L.bnh_piop_prove.restype = C.c_int
        L.bnh_piop_prove.argtypes = [
            C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p),
            C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(F128), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
            C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(F128), C.c_uint32, C.POINTER(F128), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32,
            C.POINTER(C.c_uint32), C.POINTER(F128), C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double),
        ]
        L.bnh_eqind_sumcheck_prove.restype = C.c_int
        L.bnh_eqind_sumcheck_prove.argtypes = [
            C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32),
            C.POINTER(C.c_uint32), C.POINTER(F128), C.POINTER(F128), C.c_void_p, C.c_uint64, C.POINTER(F128), C.POINTER(F128), C.POINTER(F128), C.POINTER(F128),
        ]
        L.bnh_rccl_open.argtypes = [C.c_char_p]
        L.bnh_rccl_unique_id.argtypes = [C.c_void_p]
        L.bnh_rccl_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.bnh_rccl_destroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class SumcheckPlan:
    """Pre-marshalled arguments of one prove so repeated runs have no per-call Python work."""

    def __init__(self, hal, n_vars, multilins, scratch, comps, sums, batch_coeff, challenges, reduce=None, d_partial=0,
                 rccl_comm=None, world=1, d_gathered=0, shm=None, tail_rounds=False, peer=False):
        """tail_rounds (shm or RCCL exchange): `challenges` holds n_vars + log2(world) values and the run
        also does the residual rounds; round_coeffs() then has n_vars + log2(world) entries.
        peer: the ranks' partial round evaluations are XORed on the devices inside the kernels' finalize step (the
        context must hold a connected PeerExchange); `shm` then only rebuilds the residual instance."""
        self.peer = bool(peer)
        self.hal = hal
        self.n_vars = n_vars
        self.m = len(multilins)
        self.ptrs = (C.c_void_p * self.m)(*[s.ptr for s in multilins])
        self.scratch = scratch
        flat = [i for pair in comps for i in pair]
        self.n_comps = len(comps)
        self.comps = (C.c_uint32 * max(1, len(flat)))(*flat)
        self.sums = _f128_array(list(sums))
        self.bc = to_f128(batch_coeff)
        self.ch = _f128_array(list(challenges))
        self.tail_rounds = bool(tail_rounds and (shm is not None or rccl_comm is not None) and world > 1)
        self.n_rounds = n_vars + ((world.bit_length() - 1) if self.tail_rounds else 0)
        assert len(challenges) >= self.n_rounds
        self.coeffs = (F128 * (3 * self.n_rounds))()
        self.final = (F128 * self.m)()
        self.reduce = REDUCE_FN(reduce) if reduce is not None else C.cast(None, REDUCE_FN)
        self.d_partial = d_partial
        self.rccl_comm, self.world, self.d_gathered = rccl_comm, world, d_gathered
        self.shm = shm

    def run(self):
        rc = host_lib().bnh_bivariate_sumcheck_prove(
            self.hal._h, self.n_vars, self.m, self.ptrs, self.scratch.ptr, self.scratch.len, self.n_comps, self.comps,
            self.sums, C.byref(self.bc), self.ch, self.coeffs, self.final, self.reduce, None, self.d_partial,
            self.rccl_comm, self.world, self.d_gathered, self.shm, (1 if self.tail_rounds else 0) | (2 if self.peer else 0),
        )
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())

    def round_coeffs(self):
        return [[from_f128(self.coeffs[3 * r + i]) for i in range(3)] for r in range(self.n_rounds)]

    def final_evals(self):
        return [from_f128(self.final[j]) for j in range(self.m)]


class MlecheckPlan:
    """One BivariateMLEcheckProver run of the compiled host mirror (bnh_bivariate_mlecheck_prove)."""

    def __init__(self, hal, n_vars, multilins, eq_ind, eq_ind_challenges, scratch, comps, sums, batch_coeff, challenges):
        self.hal, self.n_vars, self.m = hal, n_vars, len(multilins)
        self.ptrs = (C.c_void_p * self.m)(*[s.ptr for s in multilins])
        self.eq_ind, self.scratch = eq_ind, scratch
        self.eqc = _f128_array(list(eq_ind_challenges))
        flat = [i for pair in comps for i in pair]
        self.n_comps = len(comps)
        self.comps = (C.c_uint32 * max(1, len(flat)))(*flat)
        self.sums = _f128_array(list(sums))
        self.bc = to_f128(batch_coeff)
        self.ch = _f128_array(list(challenges))
        self.coeffs = (F128 * (4 * n_vars))()
        self.final = (F128 * (self.m + 1))()

    def run(self):
        rc = host_lib().bnh_bivariate_mlecheck_prove(
            self.hal._h, self.n_vars, self.m, self.ptrs, self.eq_ind.ptr, self.eqc, self.scratch.ptr, self.scratch.len,
            self.n_comps, self.comps, self.sums, C.byref(self.bc), self.ch, self.coeffs, self.final,
        )
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())

    def last_mode(self):
        """1: WeightedMLEcheckProver ran; 0: the literal BivariateMLEcheckProver mirror."""
        return host_lib().bnh_mlecheck_last_mode()

    def round_coeffs(self):
        return [[from_f128(self.coeffs[4 * r + i]) for i in range(4)] for r in range(self.n_vars)]

    def final_evals(self):
        return [from_f128(self.final[j]) for j in range(self.m + 1)]


class MlecheckProver:
    """The MLE-check prover behind its handle (bnh_mlecheck_*): SumcheckProver::{execute, fold, finish} one call each,
    challenges supplied round by round."""

    def __init__(self, hal, n_vars, multilins, eq_ind, eq_ind_challenges, scratch, comps, sums):
        self.m = len(multilins)
        ptrs = (C.c_void_p * self.m)(*[s.ptr for s in multilins])
        flat = [i for pair in comps for i in pair]
        cc = (C.c_uint32 * max(1, len(flat)))(*flat)
        self._h = C.c_void_p()
        self._keep = (multilins, eq_ind, scratch)
        rc = host_lib().bnh_mlecheck_new(hal._h, n_vars, self.m, ptrs, eq_ind.ptr, _f128_array(list(eq_ind_challenges)), scratch.ptr, scratch.len,
                                         len(comps), cc, _f128_array(list(sums)), C.byref(self._h))
        self._check(rc)
        self.mode = host_lib().bnh_mlecheck_last_mode()

    @staticmethod
    def _check(rc):
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())

    def execute(self, batch_coeff):
        bc, out = to_f128(batch_coeff), (F128 * 4)()
        self._check(host_lib().bnh_mlecheck_execute(self._h, C.byref(bc), out))
        return [from_f128(out[i]) for i in range(4)]

    def fold(self, challenge):
        z = to_f128(challenge)
        self._check(host_lib().bnh_mlecheck_fold(self._h, C.byref(z)))

    def finish(self):
        out = (F128 * (self.m + 1))()
        self._check(host_lib().bnh_mlecheck_finish(self._h, out))
        return [from_f128(out[j]) for j in range(self.m + 1)]

    def close(self):
        if self._h:
            host_lib().bnh_mlecheck_free(self._h)
            self._h = C.c_void_p()


class FRIParams:
    """Parameters of a FRI instance as bnh_fri_commit_fold takes them (the arithmetic of FRIParams,
    crates/core/src/protocols/fri/common.rs:84-190; validation happens in the C++ mirror, binius_amd/host/fri.hpp)."""

    def __init__(self, log_dim, log_inv_rate, log_batch_size, fold_arities, n_test_queries):
        self.log_dim, self.log_inv_rate, self.log_batch_size = log_dim, log_inv_rate, log_batch_size
        self.fold_arities, self.n_test_queries = list(fold_arities), n_test_queries

    def rs_log_len(self):
        return self.log_dim + self.log_inv_rate

    def n_fold_rounds(self):
        return self.log_dim + self.log_batch_size

    def n_final_challenges(self):
        return self.n_fold_rounds() - sum(self.fold_arities)


class FriPlan:
    """FRI commit phase + every fold round + finalize through the compiled C++ mirror (bnh_fri_commit_fold,
    binius_amd/host/fri.hpp).  `scratch` takes the codeword, the folded codewords and the Merkle trees."""

    def __init__(self, hal, params, message, scratch, challenges):
        import numpy as np

        self.hal, self.p, self.message, self.scratch = hal, params, message, scratch
        self.arities = (C.c_uint32 * max(1, len(params.fold_arities)))(*params.fold_arities)
        self.ch = _f128_array(list(challenges))
        self.roots = np.zeros((len(params.fold_arities) + 1, 32), dtype=np.uint8)
        self.terminate = np.zeros((1 << (params.log_inv_rate + params.n_final_challenges()), 2), dtype=np.uint64)
        self.phase_ms = (C.c_double * 2)()

    def run(self):
        p = self.p
        rc = host_lib().bnh_fri_commit_fold(
            self.hal._h, p.log_dim, p.log_inv_rate, p.log_batch_size, self.arities, len(p.fold_arities), p.n_test_queries,
            self.message.ptr, self.scratch.ptr, self.scratch.len, self.ch, self.roots.ctypes.data, self.terminate.ctypes.data, self.phase_ms,
        )
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())
        return self.phase_ms[0], self.phase_ms[1]


class BatchSumcheckPlan:
    """p BivariateSumcheckProvers front-loaded on ONE layer (bnh_batch_sumcheck_prove = SumcheckBatchProver::run,
    protocols/sumcheck/prove/front_loaded.rs:33-203): per round execute() on every live prover, one challenge, fold() on
    every live prover.  provers: list of (n_vars, multilins, comps, sums), ascending by n_vars."""

    def __init__(self, hal, provers, scratch, batch_coeffs, challenges):
        self.hal, self.scratch = hal, scratch
        self.n = len(provers)
        desc, ptrs, flat, sums = [], [], [], []
        for n_vars, mls, comps, sm in provers:
            desc += [n_vars, len(mls), len(comps)]
            ptrs += [x.ptr for x in mls]
            flat += [i for pair in comps for i in pair]
            sums += list(sm)
        self._keep = provers
        self.desc = (C.c_uint32 * max(1, len(desc)))(*desc)
        self.ptrs = (C.c_void_p * max(1, len(ptrs)))(*ptrs)
        self.comps = (C.c_uint32 * max(1, len(flat)))(*flat)
        self.sums = _f128_array(sums if sums else [0])
        self.bcs = _f128_array(list(batch_coeffs))
        self.rounds = max([pv[0] for pv in provers]) if provers else 0
        assert len(challenges) >= self.rounds
        self.ch = _f128_array(list(challenges) if challenges else [0])
        self.proofs = (F128 * max(1, 2 * self.rounds))()
        self.total_m = len(ptrs)
        self.final = (F128 * max(1, self.total_m))()
        self.m_by_prover = [len(pv[1]) for pv in provers]

    def run(self):
        rc = host_lib().bnh_batch_sumcheck_prove(self.hal._h, self.n, self.desc, self.ptrs, self.comps, self.sums, self.scratch.ptr, self.scratch.len,
                                                 self.bcs, self.ch, self.proofs, self.final)
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())

    def round_proofs(self):
        return [[from_f128(self.proofs[2 * r]), from_f128(self.proofs[2 * r + 1])] for r in range(self.rounds)]

    def final_evals(self):
        out, at = [], 0
        for m in self.m_by_prover:
            out.append([from_f128(self.final[at + j]) for j in range(m)])
            at += m
        return out


class PiopPlan:
    """piop::prove through the compiled C++ mirror (bnh_piop_prove, binius_amd/host/piop.hpp): commit_interleaved of the merged
    message, one BivariateSumcheckProver per size, the front-loaded batch interleaved with the FRI folder
    (crates/core/src/piop/prove.rs:148-395).  committed / transparents: lists of (n_vars, device slice), ascending by n_vars;
    claims: list of (n_vars, committed index, transparent index, sum); params: FRIParams with log_dim + log_batch_size =
    total_vars; message: device slice of 2^total_vars elements (merge_multilins of the committed multilinears)."""

    KINDS = ("round_proof", "multilinear_evals", "fri_commitment", "fri_terminate")

    def __init__(self, hal, committed, transparents, claims, params, message, scratch, batch_coeffs, challenges):
        import numpy as np

        self.hal, self.p, self.message, self.scratch = hal, params, message, scratch
        self._keep = (committed, transparents)
        self.nc, self.nt = len(committed), len(transparents)
        self.c_nv = (C.c_uint32 * max(1, self.nc))(*[v for v, _ in committed])
        self.c_ptr = (C.c_void_p * max(1, self.nc))(*[s.ptr for _, s in committed])
        self.t_nv = (C.c_uint32 * max(1, self.nt))(*[v for v, _ in transparents])
        self.t_ptr = (C.c_void_p * max(1, self.nt))(*[s.ptr for _, s in transparents])
        flat = [x for c in claims for x in c[:3]]
        self.n_claims = len(claims)
        self.claims = (C.c_uint32 * max(1, len(flat)))(*flat)
        self.sums = _f128_array([c[3] for c in claims] if claims else [0])
        self.arities = (C.c_uint32 * max(1, len(params.fold_arities)))(*params.fold_arities)
        self.bcs, self.n_bcs = _f128_array(list(batch_coeffs) if batch_coeffs else [0]), len(batch_coeffs)
        self.ch, self.n_ch = _f128_array(list(challenges)), len(challenges)
        rounds = params.n_fold_rounds()
        self.max_items = 2 * rounds + self.nc + len(params.fold_arities) + 8
        self.max_scalars = 2 * rounds + self.nc + self.nt + (1 << (params.log_inv_rate + params.n_final_challenges())) + 8
        self.max_digests = len(params.fold_arities) + 1
        self.items = (C.c_uint32 * (2 * self.max_items))()
        self.scalars = (F128 * self.max_scalars)()
        self.digests = np.zeros((self.max_digests, 32), dtype=np.uint8)
        self.commitment = np.zeros(32, dtype=np.uint8)
        self.n_items, self.n_scalars, self.n_digests = C.c_uint32(), C.c_uint64(), C.c_uint32()
        self.phase_ms = (C.c_double * 2)()

    def run(self):
        p = self.p
        rc = host_lib().bnh_piop_prove(
            self.hal._h, self.nc, self.c_nv, self.c_ptr, self.nt, self.t_nv, self.t_ptr, self.n_claims, self.claims, self.sums,
            p.log_dim, p.log_inv_rate, p.log_batch_size, self.arities, len(p.fold_arities), p.n_test_queries, self.message.ptr,
            self.scratch.ptr, self.scratch.len, self.bcs, self.n_bcs, self.ch, self.n_ch, self.commitment.ctypes.data, self.items, self.max_items,
            C.byref(self.n_items), self.scalars, self.max_scalars, C.byref(self.n_scalars), self.digests.ctypes.data, self.max_digests,
            C.byref(self.n_digests), self.phase_ms,
        )
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())
        return self.phase_ms[0], self.phase_ms[1]

    def transcript(self):
        """[(kind, payload)] in writing order: lists of ints for scalar items, 32 bytes for a FRI commitment."""
        out, at_s, at_d = [], 0, 0
        for i in range(self.n_items.value):
            kind, cnt = self.KINDS[self.items[2 * i]], self.items[2 * i + 1]
            if kind == "fri_commitment":
                out.append((kind, bytes(self.digests[at_d])))
                at_d += 1
            else:
                out.append((kind, [from_f128(self.scalars[at_s + j]) for j in range(cnt)]))
                at_s += cnt
        return out


class EqIndPlan:
    """EqIndSumcheckProver over the old HAL (bnh_eqind_sumcheck_prove = binius_amd/host/eq_ind.hpp;
    crates/core/src/protocols/sumcheck/prove/eq_ind.rs:378-644): the zerocheck of a constraint set, one composition (degree 1 .. 8: `degrees`, default all 2)
    per constraint over ALL multilinears, High-to-Low.  multilins: device slices of 2^n_vars elements, FOLDED IN PLACE by run();
    compositions: list of (steps, steps_of_the_leading_form) in compile_expr's notation; sums: one claimed sum per composition;
    eq_scratch: device slice of >= 2^(n_vars - 1) elements."""

    def __init__(self, hal, n_vars, multilins, compositions, sums, eq_ind_challenges, eq_scratch, batch_coeff, challenges, degrees=None):
        from ._ffi import make_steps

        self.degrees = (C.c_uint32 * max(1, len(compositions)))(*(degrees if degrees is not None else [2] * len(compositions)))
        self.hal, self.n_vars, self.m = hal, n_vars, len(multilins)
        self._keep = (multilins, eq_scratch)
        self.ptrs = (C.c_void_p * max(1, self.m))(*[x.ptr for x in multilins])
        self.n_comps = len(compositions)
        flat = [st for c, _ in compositions for st in c]
        flat_inf = [st for _, ci in compositions for st in ci]
        self.steps, self.steps_inf = make_steps(flat) if flat else None, make_steps(flat_inf) if flat_inf else None
        self.n_steps = (C.c_uint32 * max(1, self.n_comps))(*[len(c) for c, _ in compositions])
        self.n_steps_inf = (C.c_uint32 * max(1, self.n_comps))(*[len(ci) for _, ci in compositions])
        self.sums = _f128_array(list(sums) if sums else [0])
        assert len(eq_ind_challenges) == n_vars and len(challenges) >= n_vars
        self.eqc, self.ch = _f128_array(list(eq_ind_challenges)), _f128_array(list(challenges))
        self.bc = to_f128(batch_coeff)
        self.eq_scratch = eq_scratch
        self.per_round = 2 + max([2] + [int(d) for d in (degrees or [])])  # the round polynomials' coefficients: the largest degree (>= 2) + 2
        self.coeffs = (F128 * (self.per_round * n_vars))()
        self.final = (F128 * (self.m + 1))()

    def run(self):
        rc = host_lib().bnh_eqind_sumcheck_prove(
            self.hal._h, self.n_vars, self.m, self.ptrs, self.n_comps, C.cast(self.steps, C.c_void_p), self.n_steps, C.cast(self.steps_inf, C.c_void_p),
            self.n_steps_inf, self.degrees, self.sums, self.eqc, self.eq_scratch.ptr, self.eq_scratch.len, C.byref(self.bc), self.ch, self.coeffs, self.final)
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())

    def round_coeffs(self):
        return [[from_f128(self.coeffs[self.per_round * r + i]) for i in range(self.per_round)] for r in range(self.n_vars)]

    def final_evals(self):
        return [from_f128(self.final[j]) for j in range(self.m + 1)]


class ShmExchange:
    """Intra-node exchange of a few 64-bit words per round through POSIX shared memory (host_capi.cpp
    bnh_shm_*): rank 0 creates the segment, the name travels over the torch.distributed group."""

    def __init__(self, dist, rank, world):
        import uuid

        L = host_lib()
        box = ["/bn_amd_%s" % uuid.uuid4().hex[:16] if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(box, src=0)
        self.name = box[0]
        self.world, self.rank = world, rank
        self.handle = C.c_void_p()
        if rank == 0:
            self._open(L, 1)
        if dist is not None:
            dist.barrier()  # the segment exists (and is zeroed) before anyone else maps it
        if rank != 0:
            self._open(L, 0)
        if dist is not None:
            dist.barrier()

    def _open(self, L, create):
        rc = L.bnh_shm_open(self.name.encode(), self.world, self.rank, create, C.byref(self.handle))
        if rc != 0:
            raise BnError(rc, L.bnh_last_error().decode())

    def allgather_words(self, words):
        """words: list of < 8 ints (64-bit).  Returns [rank][i]."""
        n = len(words)
        src = (C.c_uint64 * n)(*words)
        dst = (C.c_uint64 * (n * self.world))()
        rc = host_lib().bnh_shm_allgather(self.handle, src, n, dst)
        if rc != 0:
            raise BnError(rc, host_lib().bnh_last_error().decode())
        return [[int(dst[w * n + i]) for i in range(n)] for w in range(self.world)]

    def all_gather_scalars(self, vals):
        """128-bit scalars (<= 3): returns [rank][i] like TorchComm.all_gather_scalars."""
        words = []
        for v in vals:
            words += [v & ((1 << 64) - 1), v >> 64]
        per = self.allgather_words(words)
        return [[r[2 * i] | (r[2 * i + 1] << 64) for i in range(len(vals))] for r in per]

    def xor_scalars(self, vals):
        out = [0] * len(vals)
        for r in self.all_gather_scalars(vals):
            for i, v in enumerate(r):
                out[i] ^= v
        return out

    def close(self):
        if self.handle:
            host_lib().bnh_shm_close(self.handle)
            self.handle = C.c_void_p()


class PeerExchange:
    """Device-resident exchange of the round evaluations (include/binius_amd.h bn_peer_*, csrc/finalize.hpp): every rank
    owns a mailbox in fine-grained device memory, hipIpc-mapped into all ranks of the node (peers on the same device on a
    one-GPU box, xGMI peers on a node); the handles travel over the torch.distributed group.  Once connected, a
    SumcheckPlan(..., peer=True) on this context has its local rounds reduced across the ranks inside the kernels."""

    def __init__(self, hal, dist, rank, world):
        from ._ffi import _check

        self.hal, self.world, self.rank = hal, world, rank
        self.created = False
        buf = C.create_string_buffer(64)
        _check(lib().bn_peer_create(hal._h, world, rank, buf))
        self.created = True
        try:
            handles = [None] * world
            dist.all_gather_object(handles, bytes(buf.raw))
            _check(lib().bn_peer_connect(hal._h, b"".join(handles)))
            ok = 1
        except Exception:  # noqa: BLE001 -- every rank has to learn that one of them failed
            ok = 0
        flags = [None] * world
        dist.all_gather_object(flags, ok)
        if not all(flags):
            self.close()
            raise BnError(3, "peer exchange: rank(s) %s could not map the peers' mailboxes" % [i for i, f in enumerate(flags) if not f])
        dist.barrier()  # nobody writes into a mailbox that its owner has not mapped and zeroed

    def rounds(self):
        from ._ffi import _check

        st = (C.c_uint64 * 2)()
        _check(lib().bn_peer_stats(self.hal._h, st))
        return int(st[0])

    def close(self):
        if self.created:
            lib().bn_peer_destroy(self.hal._h)
            self.created = False


class RcclComm:
    """An RCCL communicator owned by the C++ host library (one per process / GPU), bootstrapped over
    an existing torch.distributed group: rank 0 creates the unique id, everyone joins."""

    def __init__(self, dist, rank, world):
        import torch

        L = host_lib()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            path = "librccl.so"
        rc = L.bnh_rccl_open(path.encode())
        if rc != 0:
            raise BnError(rc, L.bnh_last_error().decode())
        uid = C.create_string_buffer(128)
        if rank == 0:
            rc = L.bnh_rccl_unique_id(uid)
            if rc != 0:
                raise BnError(rc, L.bnh_last_error().decode())
        box = [bytes(uid.raw)]
        dist.broadcast_object_list(box, src=0)
        uid = C.create_string_buffer(box[0], 128)
        self.handle = C.c_void_p()
        rc = L.bnh_rccl_init(uid, world, rank, C.byref(self.handle))
        if rc != 0:
            raise BnError(rc, L.bnh_last_error().decode())

    def destroy(self):
        if self.handle:
            host_lib().bnh_rccl_destroy(self.handle)
            self.handle = None
