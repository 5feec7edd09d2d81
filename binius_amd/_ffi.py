"""ctypes binding of include/binius_amd.h + Python mirrors of the reference's handle types.

Mirrors (names and argument meaning follow the reference so the parity tests read like
crates/compute_test_utils/src/layer.rs):

  DevSlice        ComputeMemory::FSlice / FSliceMut handle (crates/compute/src/memory.rs:69-234),
                  ALIGNMENT = 1: (device pointer, length); slice / split_at / split_half are O(1)
                  host arithmetic, no device calls.
  BumpAllocator   crates/compute/src/alloc.rs:31-105
  Context         ComputeLayer + ComputeLayerExecutor (crates/compute/src/layer.rs:22, 100)
  KernelExec      recording KernelExecutor (layer.rs:518-590): the kernel-spec closure is run once
                  and its ops are handed to bn_kernel_launch.
"""
import ctypes as C
import os

import numpy as np

BN_OK, BN_ERR_INPUT_VALIDATION, BN_ERR_ALLOC, BN_ERR_DEVICE, BN_ERR_CORE_LIB = range(5)
_ERR_NAMES = {1: "InputValidation", 2: "Alloc", 3: "DeviceError", 4: "CoreLibError"}

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbinius_amd.so")
MASK64 = (1 << 64) - 1
ORDER_LOW_TO_HIGH, ORDER_HIGH_TO_LOW = 0, 1  # EvaluationOrder (crates/math/src/fold.rs)
HAL_ML_FOLDED, HAL_ML_TRANSPARENT = 0, 1
NTT_MAX_DIM = 64


def lib_path():
    return _SO


class BnError(RuntimeError):
    """binius_compute::Error (crates/compute/src/layer.rs:706-716)."""

    def __init__(self, code, msg):
        super().__init__("%s: %s" % (_ERR_NAMES.get(code, code), msg))
        self.code = code
        self.kind = _ERR_NAMES.get(code, str(code))


class F128(C.Structure):
    _fields_ = [("lo", C.c_uint64), ("hi", C.c_uint64)]


class Step(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint64), ("cst", F128)]


class MemMap(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("log_min_chunk_size", C.c_uint32),
        ("d_data", C.c_void_p),
        ("len", C.c_uint64),
        ("log_size", C.c_uint32),
    ]


class HalMultilinear(C.Structure):
    """bn_hal_multilinear: SumcheckMultilinear::{Folded, Transparent} (crates/hal/src/common.rs)."""

    _fields_ = [
        ("kind", C.c_uint32),
        ("tower_level", C.c_uint32),
        ("d_evals", C.c_void_p),
        ("len", C.c_uint64),
        ("suffix_eval", F128),
        ("n_vars_ml", C.c_uint32),
    ]


class HalEvaluator(C.Structure):
    """bn_hal_evaluator: what a SumcheckEvaluator contributes to one round (crates/hal/src/sumcheck_evaluator.rs)."""

    _fields_ = [
        ("composition", C.c_void_p),
        ("composition_at_infinity", C.c_void_p),
        ("eval_point_start", C.c_uint32),
        ("eval_point_end", C.c_uint32),
        ("d_eq_ind", C.c_void_p),
    ]


class KSlice(C.Structure):
    _fields_ = [("buf", C.c_uint32), ("off", C.c_uint64), ("len", C.c_uint64)]


class KOp(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("value", C.c_uint32),
        ("scalar", F128),
        ("expr", C.c_void_p),
        ("n_rows", C.c_uint32),
        ("rows", C.POINTER(KSlice)),
        ("src1", KSlice),
        ("src2", KSlice),
        ("dst", KSlice),
    ]


STEP_KINDS = {"add": 0, "mul": 1, "pow": 2, "const": 3, "var": 4}
MAP_CHUNKED, MAP_CHUNKED_MUT, MAP_LOCAL = range(3)
KOP_DECL_VALUE, KOP_SUM_COMPOSITION, KOP_ADD, KOP_ADD_ASSIGN = range(4)

_lib = None


def lib():
    """Load libbinius_amd.so.  Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            "binius_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % _SO
        )
    L = C.CDLL(_SO)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    PF = C.POINTER(F128)
    sig = {
        "bn_ctx_create": [i32, u64, C.POINTER(vp)],
        "bn_ctx_destroy": [vp],
        "bn_arena_base": [vp, C.POINTER(vp), C.POINTER(u64)],
        "bn_ctx_set_stream": [vp, vp],
        "bn_sync": [vp],
        "bn_ctx_get_stream": [vp, C.POINTER(vp)],
        "bn_copy_h2d": [vp, vp, u64, vp, u64],
        "bn_copy_d2h": [vp, vp, u64, vp, u64],
        "bn_copy_d2d": [vp, vp, u64, vp, u64],
        "bn_fill": [vp, vp, u64, PF],
        "bn_expr_compile": [vp, C.POINTER(Step), u64, C.POINTER(vp)],
        "bn_expr_free": [vp],
        "bn_expr_n_vars": [vp, C.POINTER(u32)],
        "bn_extrapolate_line": [vp, vp, u64, vp, u64, PF],
        "bn_extrapolate_line_batch": [vp, C.POINTER(vp), C.POINTER(vp), u32, u64, PF],
        "bn_extrapolate_line_batch_scaled": [vp, C.POINTER(vp), C.POINTER(vp), u32, u64, PF, u32, PF],
        "bn_tensor_expand": [vp, vp, u64, u32, PF, u32],
        "bn_inner_product": [vp, vp, u64, u32, vp, u64, PF],
        "bn_fold_left": [vp, vp, u64, u32, vp, u64, vp, u64],
        "bn_fold_right": [vp, vp, u64, u32, vp, u64, vp, u64],
        "bn_fri_fold": [vp, C.POINTER(u64), u32, u32, u32, u32, PF, u32, vp, u64, vp, u64],
        "bn_compute_composite": [vp, C.POINTER(vp), u32, u64, vp, u64, vp],
        "bn_pairwise_product_reduce": [vp, vp, u64, C.POINTER(vp), C.POINTER(u64), u32],
        "bn_log_chunks_range": [C.POINTER(MemMap), u32, C.POINTER(u32), C.POINTER(u32)],
        "bn_pick_log_chunks": [C.POINTER(MemMap), u32, C.POINTER(u32)],
        "bn_kernel_launch": [vp, C.POINTER(MemMap), u32, C.POINTER(KOp), u32, C.POINTER(u32), u32, u32, PF, vp],
        "bn_hal_round_evals": [vp, u32, u32, vp, u32, C.POINTER(HalMultilinear), u32, C.POINTER(HalEvaluator), u32, PF, u32, PF],
        "bn_hal_fold_multilinear": [vp, u32, u32, C.POINTER(HalMultilinear), PF, vp, u32, vp, u64, C.POINTER(u64)],
        "bn_ntt_forward": [vp, vp, u32, u32, C.POINTER(u64), u32, u32, u32, u32, u64, u32, u32],
        "bn_ntt_inverse": [vp, vp, u32, u32, C.POINTER(u64), u32, u32, u32, u32, u64, u32, u32],
        "bn_ntt_s_evals": [u32, u32, C.POINTER(u64)],
        "bn_scalar_mul": [PF, PF, PF],
        "bn_scalar_invert": [PF, PF],
        "bn_prof_begin": [vp],
        "bn_prof_end": [vp, C.POINTER(C.c_double), C.POINTER(u64)],
        "bn_arm_counters": [vp, C.POINTER(u64)],
        "bn_group_counters": [vp, C.POINTER(u64)],
        "bn_xor_reduce": [vp, vp, u32, u32, PF],
        "bn_host_scratch": [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)],
        "bn_device_numa_node": [C.c_int, C.POINTER(C.c_int)],
        "bn_merkle_build": [vp, vp, u64, u64, vp],
        "bn_groestl256_leaves": [vp, vp, u64, u64, vp],
        "bn_groestl256_compress_layer": [vp, vp, u64, vp],
        "bn_gather_d2h": [vp, vp, C.POINTER(u64), u64, u64, vp],
        "bn_timer_begin": [vp],
        "bn_timer_end_ms": [vp, C.POINTER(C.c_float)],
        "bn_peer_create": [vp, u32, u32, C.c_char_p],
        "bn_peer_connect": [vp, C.c_char_p],
        "bn_peer_set_active": [vp, C.c_int],
        "bn_peer_stats": [vp, C.POINTER(u64)],
        "bn_host_tail_allow_peer": [vp, C.c_int],
        "bn_host_tail_active": [vp, C.POINTER(C.c_int)],
        "bn_peer_destroy": [vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)  # AttributeError here == the .so does not export a declared symbol
        fn.argtypes = args
        fn.restype = C.c_int
    L.bn_last_error.restype = C.c_char_p
    L.bn_version.restype = C.c_char_p
    _lib = L
    return L


ABI_SYMBOLS = [
    "bn_last_error", "bn_version", "bn_ctx_create", "bn_ctx_destroy", "bn_arena_base", "bn_ctx_set_stream", "bn_sync", "bn_ctx_get_stream",
    "bn_copy_h2d", "bn_copy_d2h", "bn_copy_d2d", "bn_fill", "bn_expr_compile", "bn_expr_free", "bn_expr_n_vars",
    "bn_extrapolate_line", "bn_extrapolate_line_batch", "bn_tensor_expand", "bn_inner_product", "bn_fold_left", "bn_fold_right", "bn_fri_fold",
    "bn_compute_composite", "bn_pairwise_product_reduce", "bn_log_chunks_range", "bn_pick_log_chunks",
    "bn_kernel_launch", "bn_ntt_forward", "bn_ntt_inverse", "bn_ntt_s_evals", "bn_scalar_mul", "bn_scalar_invert",
    "bn_timer_begin", "bn_timer_end_ms", "bn_prof_begin", "bn_prof_end", "bn_arm_counters", "bn_group_counters", "bn_xor_reduce", "bn_host_scratch", "bn_device_numa_node",
    "bn_merkle_build", "bn_groestl256_leaves", "bn_groestl256_compress_layer", "bn_gather_d2h",
    "bn_hal_round_evals", "bn_hal_fold_multilinear", "bn_extrapolate_line_batch_scaled",
    "bn_peer_create", "bn_peer_connect", "bn_peer_set_active", "bn_peer_stats", "bn_peer_destroy", "bn_host_tail_allow_peer", "bn_host_tail_active",
]


def _check(rc):
    if rc != BN_OK:
        raise BnError(rc, lib().bn_last_error().decode())


def device_numa_node(device=0):
    """NUMA node of the host the device hangs off, or None if the platform does not say (bn_device_numa_node)."""
    node = C.c_int(-1)
    _check(lib().bn_device_numa_node(int(device), C.byref(node)))
    return node.value if node.value >= 0 else None


def bind_host_thread_to_device(device=0):
    """Restrict the calling thread to the CPUs of the device's NUMA node (what `numactl --cpunodebind` does): every small
    sumcheck round is a host -> device -> host round trip, and from the other socket of a 2-socket host each one also
    crosses the socket interconnect (15.0 -> 17.1 us per two-round launch measured).  Returns a description of what was
    done; never widens the current affinity, does nothing when the node or its CPU list is unknown."""
    import os

    try:
        node = device_numa_node(device)
    except Exception as ex:  # (a measurement convenience must never be the reason a run fails)
        return "unchanged (%s)" % type(ex).__name__
    if node is None or not hasattr(os, "sched_setaffinity"):
        return "unchanged (no NUMA information for the device)"
    try:
        txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    except OSError:
        return "unchanged (node %d has no cpulist)" % node
    cpus = set()
    try:
        for part in txt.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
    except ValueError:
        return "unchanged (unreadable cpulist of node %d)" % node
    try:
        want = cpus & os.sched_getaffinity(0)
        if not want:
            return "unchanged (none of node %d's CPUs are allowed to this process)" % node
        os.sched_setaffinity(0, want)
    except OSError as ex:
        return "unchanged (sched_setaffinity: %s)" % ex
    return "NUMA node %d (%d CPUs)" % (node, len(want))


def to_f128(x):
    x = int(x)
    return F128(x & MASK64, (x >> 64) & MASK64)


def from_f128(f):
    return int(f.lo) | (int(f.hi) << 64)


def _f128_array(vals):
    arr = (F128 * max(1, len(vals)))()
    for i, v in enumerate(vals):
        arr[i] = to_f128(v)
    return arr


def make_steps(steps):
    out = (Step * max(1, len(steps)))()
    for i, s in enumerate(steps):
        k = STEP_KINDS[s[0]]
        out[i].kind = k
        if k in (0, 1, 2):
            out[i].a, out[i].b = s[1], s[2]
        elif k == 3:
            out[i].cst = to_f128(s[1])
        else:
            out[i].a = s[1]
    return out


class HostField:
    """O(1)-per-round protocol scalars on the host (bn_scalar_mul / bn_scalar_invert)."""

    @staticmethod
    def mul(a, b):
        x, y, o = to_f128(a), to_f128(b), F128()
        _check(lib().bn_scalar_mul(C.byref(x), C.byref(y), C.byref(o)))
        return from_f128(o)

    @staticmethod
    def invert(a):
        x, o = to_f128(a), F128()
        _check(lib().bn_scalar_invert(C.byref(x), C.byref(o)))
        return from_f128(o)


# ------------------------------------------------------------------ memory handles
class DevSlice:
    """Opaque handle to a slice of F in device memory: (pointer, len).  ALIGNMENT = 1."""

    ALIGNMENT = 1
    __slots__ = ("ptr", "len")

    def __init__(self, ptr, length):
        self.ptr = int(ptr)
        self.len = int(length)

    def __len__(self):
        return self.len

    def __repr__(self):
        return "DevSlice(0x%x, len=%d)" % (self.ptr, self.len)

    def slice(self, start=0, end=None):
        end = self.len if end is None else end
        assert 0 <= start <= end <= self.len, "slice out of range"
        return DevSlice(self.ptr + 16 * start, end - start)

    def split_at(self, mid):
        return self.slice(0, mid), self.slice(mid, self.len)

    def split_half(self):
        assert self.len > 1 and (self.len & (self.len - 1)) == 0, "data length must be a power of two greater than 1"
        return self.split_at(self.len // 2)

    def chunks(self, chunk_len):
        assert self.len % chunk_len == 0
        return [self.slice(i, i + chunk_len) for i in range(0, self.len, chunk_len)]


class BumpAllocator:
    """crates/compute/src/alloc.rs:31-105 over a DevSlice (or another allocator's remainder)."""

    def __init__(self, buffer):
        self._buf = buffer

    def alloc(self, n):
        if self._buf.len < n:
            raise BnError(BN_ERR_ALLOC, "allocator is out of memory")
        lhs, rhs = self._buf.split_at(n)
        self._buf = rhs
        return lhs

    def capacity(self):
        return self._buf.len

    def subscope_allocator(self):
        return BumpAllocator(self._buf.slice())


class Expr:
    """ExprEval: handle returned by compile_expr (layer.rs:57)."""

    def __init__(self, handle, steps):
        self.handle = handle
        self.steps = list(steps)

    def n_vars(self):
        n = C.c_uint32()
        _check(lib().bn_expr_n_vars(self.handle, C.byref(n)))
        return n.value

    def free(self):
        if self.handle:
            lib().bn_expr_free(self.handle)
            self.handle = None


# ------------------------------------------------------------------ recording KernelExecutor
class KernelBuffer:
    """KernelBuffer::{Ref,Mut} (layer.rs:682-704): a chunk-relative view of mapped buffer `buf`."""

    __slots__ = ("buf", "off", "len", "mutable")

    def __init__(self, buf, off, length, mutable):
        self.buf, self.off, self.len, self.mutable = buf, off, length, mutable

    def __len__(self):
        return self.len

    def to_ref(self):
        return KernelBuffer(self.buf, self.off, self.len, False)

    def slice(self, start=0, end=None):
        end = self.len if end is None else end
        assert 0 <= start <= end <= self.len
        return KernelBuffer(self.buf, self.off + start, end - start, self.mutable)

    def triple(self):
        return (self.buf, self.off, self.len)


class KernelValue:
    __slots__ = ("id",)

    def __init__(self, vid):
        self.id = vid


class KernelExec:
    """Recording KernelExecutor: collects the ops the kernel-spec closure issues."""

    def __init__(self):
        self.ops = []
        self.n_values = 0

    def decl_value(self, init):
        v = KernelValue(self.n_values)
        self.n_values += 1
        self.ops.append({"op": "decl", "value": v.id, "init": int(init)})
        return v

    def sum_composition_evals(self, inputs, composition, batch_coeff, accumulator):
        row_len = len(inputs[0]) if inputs else 0
        for r in inputs:
            assert len(r) == row_len  # SlicesBatch::new (memory.rs:39-45)
        self.ops.append(
            {"op": "sum", "value": accumulator.id, "expr": composition, "coeff": int(batch_coeff), "rows": [r.triple() for r in inputs]}
        )

    def add(self, log_len, src1, src2, dst):
        assert len(src1) == 1 << log_len and len(src2) == 1 << log_len and len(dst) == 1 << log_len
        assert dst.mutable
        self.ops.append({"op": "add", "src1": src1.triple(), "src2": src2.triple(), "dst": dst.triple()})

    def add_assign(self, log_len, src, dst):
        assert len(src) == 1 << log_len and len(dst) == 1 << log_len
        assert dst.mutable
        self.ops.append({"op": "add_assign", "src": src.triple(), "dst": dst.triple()})


def _make_maps(mem_maps):
    mm = (MemMap * max(1, len(mem_maps)))()
    for i, m in enumerate(mem_maps):
        if m[0] == "local":
            mm[i].kind = MAP_LOCAL
            mm[i].log_size = m[1]
        else:
            mm[i].kind = MAP_CHUNKED if m[0] == "chunked" else MAP_CHUNKED_MUT
            mm[i].d_data = m[1].ptr
            mm[i].len = m[1].len
            mm[i].log_min_chunk_size = m[2]
    return mm


def log_chunks_range(mem_maps):
    """KernelMemMap::log_chunks_range (layer.rs:617-644). Host only."""
    mm = _make_maps(mem_maps)
    s, e = C.c_uint32(), C.c_uint32()
    _check(lib().bn_log_chunks_range(mm, len(mem_maps), C.byref(s), C.byref(e)))
    return range(s.value, e.value)


def ntt_s_evals(tw_level, log_domain):
    """OnTheFlyTwiddleAccess::generate for the canonical subspace. Host only."""
    s = np.zeros(NTT_MAX_DIM * NTT_MAX_DIM, dtype=np.uint64)
    _check(lib().bn_ntt_s_evals(tw_level, log_domain, s.ctypes.data_as(C.POINTER(C.c_uint64))))
    return s


# ------------------------------------------------------------------ the layer
class Context:
    """ComputeLayer + executor over one GPU.  `arena_elems` F elements of device memory are
    allocated up front (like FastCpuLayerHolder::new(host, dev)); `dev_alloc()` bump-allocates."""

    def __init__(self, device=0, arena_elems=0):
        self._h = C.c_void_p()
        _check(lib().bn_ctx_create(device, arena_elems, C.byref(self._h)))
        self.device = device
        base, n = C.c_void_p(), C.c_uint64()
        _check(lib().bn_arena_base(self._h, C.byref(base), C.byref(n)))
        self.arena = DevSlice(base.value or 0, n.value)
        self._exprs = []

    def close(self):
        if self._h:
            for e in self._exprs:
                e.free()
            lib().bn_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def dev_alloc(self):
        return BumpAllocator(self.arena.slice())

    def set_stream(self, hip_stream):
        _check(lib().bn_ctx_set_stream(self._h, hip_stream))

    def sync(self):
        _check(lib().bn_sync(self._h))

    def timer_begin(self):
        _check(lib().bn_timer_begin(self._h))

    def timer_end_ms(self):
        ms = C.c_float()
        _check(lib().bn_timer_end_ms(self._h, C.byref(ms)))
        return ms.value

    PROF_CLASSES = ("round_eval", "fold", "tensor_expand", "ntt", "other", "fold_eval", "tail", "fold_eval_small", "fold_eval_mfma", "round_eval_mfma",
                    "fold_eval8")

    def prof_begin(self):
        _check(lib().bn_prof_begin(self._h))

    def prof_end(self):
        ms = (C.c_double * len(self.PROF_CLASSES))()
        cnt = (C.c_uint64 * len(self.PROF_CLASSES))()
        _check(lib().bn_prof_end(self._h, ms, cnt))
        return {k: (ms[i], int(cnt[i])) for i, k in enumerate(self.PROF_CLASSES)}

    def arm_counters(self):
        """Armed rounds (csrc/arm.hpp): {hits, cancels, expired} since the context was created; two-round launches
        (csrc/kernels_foldeval8.hip): their number and the rounds the host answered from their precomputed sums."""
        c = (C.c_uint64 * 15)()
        _check(lib().bn_arm_counters(self._h, c))
        return {"hits": int(c[0]), "cancels": int(c[1]), "expired": int(c[2]), "ns_wait": int(c[3]), "ns_launch": int(c[4]), "ns_parse": int(c[5]),
                "hosted": int(c[6]), "two_round": int(c[7]), "shadow_created": int(c[8]), "shadow_rounds": int(c[9]), "shadow_dropped": int(c[10]),
                # host tail (abi_kernels.cpp): instances the host took over, round evaluations it answered, fold chains launched
                "ht_started": int(c[11]), "ht_rounds": int(c[12]), "ht_flushed": int(c[13]), "ht_max": int(c[14])}

    def group_counters(self):
        """Claim groups (csrc/abi_group.cpp): how the call shape of piop::prove -- k product claims over m multilinears per
        prover, several provers front-loaded on one layer -- was run: launches of the group kernel, claims evaluated fused
        with their folds / on pre-folded arrays, plain fold launches of a round, claims of other provers carried along,
        execute() calls answered from sums computed ahead, execute() calls on this path, deferred folds forced out."""
        c = (C.c_uint64 * 14)()
        _check(lib().bn_group_counters(self._h, c))
        keys = ("launches", "jobs_fused", "jobs_eval", "prefolds", "spec_jobs", "spec_hits", "evals", "flushed_folds", "hosted_started", "hosted_evals",
                "hosted_folds", "hosted_writebacks", "jobs_fold", "chains")
        return {k: int(c[i]) for i, k in enumerate(keys)}

    # ---- ComputeLayer
    def copy_h2d(self, src, dst):
        """src: numpy uint64 array of shape (n, 2)."""
        src = np.ascontiguousarray(src, dtype=np.uint64).reshape(-1, 2)
        _check(lib().bn_copy_h2d(self._h, src.ctypes.data, src.shape[0], dst.ptr, dst.len))

    def copy_d2h(self, src, dst=None):
        if dst is None:
            dst = np.zeros((src.len, 2), dtype=np.uint64)
        assert dst.dtype == np.uint64 and dst.flags["C_CONTIGUOUS"]
        _check(lib().bn_copy_d2h(self._h, src.ptr, src.len, dst.ctypes.data, dst.reshape(-1, 2).shape[0]))
        return dst

    def copy_d2d(self, src, dst):
        _check(lib().bn_copy_d2d(self._h, src.ptr, src.len, dst.ptr, dst.len))

    def fill(self, slice_, value):
        v = to_f128(value)
        _check(lib().bn_fill(self._h, slice_.ptr, slice_.len, C.byref(v)))

    def compile_expr(self, steps):
        st = make_steps(steps)
        h = C.c_void_p()
        _check(lib().bn_expr_compile(self._h, st, len(steps), C.byref(h)))
        e = Expr(h, steps)
        self._exprs.append(e)
        return e

    # ---- ComputeLayerExecutor
    def extrapolate_line(self, evals_0, evals_1, z):
        zz = to_f128(z)
        _check(lib().bn_extrapolate_line(self._h, evals_0.ptr, evals_0.len, evals_1.ptr, evals_1.len, C.byref(zz)))

    def extrapolate_line_batch(self, evals_0, evals_1, z):
        """The `map` scope of a fold (v3/bivariate_product.rs:217-228): the same z applied to several
        (evals_0, evals_1) pairs of equal length."""
        n = evals_0[0].len if evals_0 else 0
        for a, b in zip(evals_0, evals_1):
            if a.len != n or b.len != n:
                raise BnError(BN_ERR_INPUT_VALIDATION, "input validation: extrapolate_line batch slices differ in length")
        p0 = (C.c_void_p * len(evals_0))(*[a.ptr for a in evals_0])
        p1 = (C.c_void_p * len(evals_1))(*[b.ptr for b in evals_1])
        zz = to_f128(z)
        _check(lib().bn_extrapolate_line_batch(self._h, p0, p1, len(evals_0), n, C.byref(zz)))

    def extrapolate_line_batch_scaled(self, evals_0, evals_1, z, scale_mask, hi_scale):
        """Extension: the batch fold, then the upper half of every folded array whose bit is set times hi_scale."""
        n = evals_0[0].len
        p0 = (C.c_void_p * len(evals_0))(*[a.ptr for a in evals_0])
        p1 = (C.c_void_p * len(evals_1))(*[b.ptr for b in evals_1])
        zz, hs = to_f128(z), to_f128(hi_scale)
        _check(lib().bn_extrapolate_line_batch_scaled(self._h, p0, p1, len(evals_0), n, C.byref(zz), scale_mask, C.byref(hs)))

    def tensor_expand(self, log_n, coordinates, data):
        c = _f128_array(list(coordinates))
        _check(lib().bn_tensor_expand(self._h, data.ptr, data.len, log_n, c, len(coordinates)))

    def inner_product(self, a_in, tower_level, b_in):
        out = F128()
        _check(lib().bn_inner_product(self._h, a_in.ptr, a_in.len, tower_level, b_in.ptr, b_in.len, C.byref(out)))
        return from_f128(out)

    def fold_left(self, mat, tower_level, vec, out):
        _check(lib().bn_fold_left(self._h, mat.ptr, mat.len, tower_level, vec.ptr, vec.len, out.ptr, out.len))

    def fold_right(self, mat, tower_level, vec, out):
        _check(lib().bn_fold_right(self._h, mat.ptr, mat.len, tower_level, vec.ptr, vec.len, out.ptr, out.len))

    # ---- the old HAL (binius_hal::ComputationBackend, crates/hal/src/backend.rs:35-84) on device-resident multilinears
    @staticmethod
    def _hal_ml(ml):
        """ml: ('folded', DevSlice, suffix_eval) | ('transparent', DevSlice of packed values, tower_level, n_vars_ml)."""
        m = HalMultilinear()
        if ml[0] == "folded":
            m.kind, m.d_evals, m.len, m.suffix_eval = HAL_ML_FOLDED, ml[1].ptr, ml[1].len, to_f128(ml[2])
        else:
            m.kind, m.d_evals, m.len, m.tower_level, m.n_vars_ml = HAL_ML_TRANSPARENT, ml[1].ptr, ml[1].len, ml[2], ml[3]
        return m

    def hal_round_evals(self, order, n_vars, tensor_query, multilinears, evaluators, nontrivial_points):
        """sumcheck_compute_round_evals (backend.rs:52-67).  evaluators: dicts {composition: Expr, composition_at_infinity:
        Expr, start, end, eq_ind: DevSlice | None}; tensor_query: DevSlice of the query expansion or None.  Returns one list
        of values per evaluator (its evaluation point indices start..end)."""
        mls = (HalMultilinear * max(1, len(multilinears)))(*[self._hal_ml(m) for m in multilinears])
        evs = (HalEvaluator * max(1, len(evaluators)))()
        total = 0
        for k, e in enumerate(evaluators):
            evs[k].composition = e["composition"].handle
            evs[k].composition_at_infinity = e["composition_at_infinity"].handle
            evs[k].eval_point_start, evs[k].eval_point_end = e["start"], e["end"]
            evs[k].d_eq_ind = e["eq_ind"].ptr if e.get("eq_ind") is not None else None
            total += max(0, e["end"] - e["start"])
        pts = _f128_array(list(nontrivial_points))
        out = (F128 * max(1, total))()
        q_ptr = tensor_query.ptr if tensor_query is not None else None
        q_vars = (tensor_query.len.bit_length() - 1) if tensor_query is not None else 0
        _check(lib().bn_hal_round_evals(self._h, order, n_vars, q_ptr, q_vars, mls, len(multilinears), evs, len(evaluators), pts,
                                        len(nontrivial_points), out))
        res, off = [], 0
        for e in evaluators:
            cnt = max(0, e["end"] - e["start"])
            res.append([from_f128(out[off + t]) for t in range(cnt)])
            off += cnt
        return res

    def hal_fold_multilinear(self, order, n_vars, multilinear, challenge, tensor_query, out):
        """One multilinear of sumcheck_fold_multilinears (backend.rs:69-78): returns the number of evaluations written."""
        m = self._hal_ml(multilinear)
        z = to_f128(challenge)
        n = C.c_uint64()
        q_ptr = tensor_query.ptr if tensor_query is not None else None
        q_vars = (tensor_query.len.bit_length() - 1) if tensor_query is not None else 0
        _check(lib().bn_hal_fold_multilinear(self._h, order, n_vars, C.byref(m), C.byref(z), q_ptr, q_vars, out.ptr, out.len, C.byref(n)))
        return n.value

    def fri_fold(self, s_evals, tw_level, log_domain, log_len, log_batch_size, challenges, data_in, data_out):
        ch = _f128_array(list(challenges))
        _check(
            lib().bn_fri_fold(
                self._h, s_evals.ctypes.data_as(C.POINTER(C.c_uint64)), tw_level, log_domain, log_len, log_batch_size,
                ch, len(challenges), data_in.ptr, data_in.len, data_out.ptr, data_out.len,
            )
        )

    # ---- Merkle commitment with Groestl-256 (include/binius_amd.h; digests are 2 arena elements each)
    def merkle_build(self, elems, batch_size, nodes):
        """nodes: DevSlice of 2 * (2 * n_leaves - 1) elements receiving the flattened tree."""
        if batch_size and elems.len % batch_size == 0 and nodes.len != 2 * (2 * (elems.len // batch_size) - 1):
            raise BnError(BN_ERR_INPUT_VALIDATION, "input validation: merkle_build: nodes must hold 2 * n_leaves - 1 digests")
        _check(lib().bn_merkle_build(self._h, elems.ptr, elems.len, batch_size, nodes.ptr))

    def groestl256_leaves(self, elems, batch_size, digests):
        if batch_size and elems.len % batch_size == 0 and digests.len != 2 * (elems.len // batch_size):
            raise BnError(BN_ERR_INPUT_VALIDATION, "input validation: groestl256_leaves: digests must hold one digest per batch")
        _check(lib().bn_groestl256_leaves(self._h, elems.ptr, elems.len, batch_size, digests.ptr))

    def groestl256_compress_layer(self, prev, nxt):
        if prev.len != 2 * nxt.len:
            raise BnError(BN_ERR_INPUT_VALIDATION, "input validation: compress_layer: the next layer is half the previous one")
        _check(lib().bn_groestl256_compress_layer(self._h, prev.ptr, nxt.len // 2, nxt.ptr))

    def gather_d2h(self, src, offsets, item_elems):
        """(len(offsets), item_elems, 2) uint64 array: src[offsets[i] : offsets[i] + item_elems] for every i."""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        if len(offs) and int(offs.max()) + item_elems > src.len:
            raise BnError(BN_ERR_INPUT_VALIDATION, "input validation: gather: item out of range")
        out = np.zeros((len(offs), item_elems, 2), dtype=np.uint64)
        _check(lib().bn_gather_d2h(self._h, src.ptr, offs.ctypes.data_as(C.POINTER(C.c_uint64)), len(offs), item_elems, out.ctypes.data))
        return out

    def compute_composite(self, inputs, output, composition):
        rows = (C.c_void_p * max(1, len(inputs)))(*[r.ptr for r in inputs])
        row_len = inputs[0].len if inputs else 0
        for r in inputs:
            assert r.len == row_len  # SlicesBatch::new
        _check(lib().bn_compute_composite(self._h, rows, len(inputs), row_len, output.ptr, output.len, composition.handle))

    def pairwise_product_reduce(self, inp, round_outputs):
        outs = (C.c_void_p * max(1, len(round_outputs)))(*[r.ptr for r in round_outputs])
        lens = (C.c_uint64 * max(1, len(round_outputs)))(*[r.len for r in round_outputs])
        _check(lib().bn_pairwise_product_reduce(self._h, inp.ptr, inp.len, outs, lens, len(round_outputs)))

    # ---- accumulate_kernels / map_kernels
    def pick_log_chunks(self, mem_maps):
        mm = _make_maps(mem_maps)
        lc = C.c_uint32()
        _check(lib().bn_pick_log_chunks(mm, len(mem_maps), C.byref(lc)))
        return lc.value

    def record(self, map_fn, mem_maps):
        """Run the kernel-spec closure once against the recording executor."""
        log_chunks = self.pick_log_chunks(mem_maps)
        buffers = []
        for i, m in enumerate(mem_maps):
            if m[0] == "local":
                buffers.append(KernelBuffer(i, 0, (1 << m[1]) >> log_chunks, True))
            else:
                buffers.append(KernelBuffer(i, 0, m[1].len >> log_chunks, m[0] == "chunked_mut"))
        ke = KernelExec()
        rets = map_fn(ke, log_chunks, buffers)
        return ke.ops, [v.id for v in (rets or [])], log_chunks

    def kernel_launch(self, mem_maps, ops, ret_ids, log_chunks, d_out=None, want_host=True):
        mm = _make_maps(mem_maps)
        kops = (KOp * max(1, len(ops)))()
        keep = []
        for i, o in enumerate(ops):
            if o["op"] == "decl":
                kops[i].kind = KOP_DECL_VALUE
                kops[i].value = o["value"]
                kops[i].scalar = to_f128(o["init"])
            elif o["op"] == "sum":
                kops[i].kind = KOP_SUM_COMPOSITION
                kops[i].value = o["value"]
                kops[i].scalar = to_f128(o["coeff"])
                kops[i].expr = o["expr"].handle
                rows = (KSlice * max(1, len(o["rows"])))(*[KSlice(*r) for r in o["rows"]])
                keep.append(rows)
                kops[i].rows = rows
                kops[i].n_rows = len(o["rows"])
            elif o["op"] == "add":
                kops[i].kind = KOP_ADD
                kops[i].src1, kops[i].src2, kops[i].dst = KSlice(*o["src1"]), KSlice(*o["src2"]), KSlice(*o["dst"])
            else:
                kops[i].kind = KOP_ADD_ASSIGN
                kops[i].src1, kops[i].dst = KSlice(*o["src"]), KSlice(*o["dst"])
        rv = (C.c_uint32 * max(1, len(ret_ids)))(*ret_ids)
        h_out = (F128 * max(1, len(ret_ids)))()
        _check(
            lib().bn_kernel_launch(
                self._h, mm, len(mem_maps), kops, len(ops), rv, len(ret_ids), log_chunks,
                h_out if (want_host and ret_ids) else None, d_out,
            )
        )
        return [from_f128(h_out[i]) for i in range(len(ret_ids))] if want_host else None

    def accumulate_kernels(self, map_fn, mem_maps):
        ops, ret_ids, log_chunks = self.record(map_fn, mem_maps)
        return self.kernel_launch(mem_maps, ops, ret_ids, log_chunks)

    def map_kernels(self, map_fn, mem_maps):
        ops, _rets, log_chunks = self.record(map_fn, mem_maps)
        self.kernel_launch(mem_maps, ops, [], log_chunks)

    # ---- AdditiveNTT
    def ntt_forward(self, data_ptr, elem_level, tw_level, s_evals, log_domain, log_x, log_y, log_z, coset=0, coset_bits=0, skip_rounds=0):
        _check(
            lib().bn_ntt_forward(
                self._h, data_ptr, elem_level, tw_level, s_evals.ctypes.data_as(C.POINTER(C.c_uint64)), log_domain,
                log_x, log_y, log_z, coset, coset_bits, skip_rounds,
            )
        )

    def ntt_inverse(self, data_ptr, elem_level, tw_level, s_evals, log_domain, log_x, log_y, log_z, coset=0, coset_bits=0, skip_rounds=0):
        _check(
            lib().bn_ntt_inverse(
                self._h, data_ptr, elem_level, tw_level, s_evals.ctypes.data_as(C.POINTER(C.c_uint64)), log_domain,
                log_x, log_y, log_z, coset, coset_bits, skip_rounds,
            )
        )

    # raw byte copies for non-F128 NTT data
    def copy_bytes_h2d(self, src_np, dst_ptr):
        n16 = (src_np.nbytes + 15) // 16
        buf = np.zeros(n16 * 2, dtype=np.uint64)
        buf.view(np.uint8)[: src_np.nbytes] = src_np.view(np.uint8).reshape(-1)
        _check(lib().bn_copy_h2d(self._h, buf.ctypes.data, n16, dst_ptr, n16))

    def copy_bytes_d2h(self, src_ptr, dst_np):
        n16 = (dst_np.nbytes + 15) // 16
        buf = np.zeros(n16 * 2, dtype=np.uint64)
        _check(lib().bn_copy_d2h(self._h, src_ptr, n16, buf.ctypes.data, n16))
        dst_np.view(np.uint8).reshape(-1)[:] = buf.view(np.uint8)[: dst_np.nbytes]
        return dst_np
