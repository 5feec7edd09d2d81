"""binius_amd -- MI355X (gfx950) compute backend for the Binius prover hot path.

This package is only the Python-side plumbing (ctypes binding of the C ABI in
``include/binius_amd.h`` plus small mirrors of the reference's ``ComputeMemory`` /
``BumpAllocator`` handle types) used by the parity tests, ``bench.py`` and
``__graft_entry__.py``.  The product is ``libbinius_amd.so``: hand-written HIP kernels behind an
``extern "C"`` boundary.  There is NO CPU fallback: if the shared library is missing or no GPU is
visible, loading / context creation fails loudly.
"""
from ._ffi import (  # noqa: F401
    BN_OK,
    BnError,
    Context,
    DevSlice,
    BumpAllocator,
    Expr,
    HostField,
    bind_host_thread_to_device,
    device_numa_node,
    lib,
    lib_path,
    log_chunks_range,
    ntt_s_evals,
)


def _ffi_ints_to_arr(vals):
    """list of Python ints -> numpy (n, 2) uint64 array in BinaryField128b memory layout."""
    import numpy as np

    a = np.zeros((len(vals), 2), dtype=np.uint64)
    for i, v in enumerate(vals):
        a[i, 0] = v & ((1 << 64) - 1)
        a[i, 1] = v >> 64
    return a
