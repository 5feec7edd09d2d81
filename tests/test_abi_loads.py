"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/binius_amd.h declares, host-only entry points work, and a missing GPU fails loudly
(no compute calls here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "binius_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bn_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def ffi():
    import __graft_entry__ as g

    g.build()
    import binius_amd._ffi as f

    return f


def test_library_exports_every_declared_symbol(ffi):
    L = ffi.lib()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), "libbinius_amd.so does not export %s" % s
    assert set(ffi.ABI_SYMBOLS) == set(syms)
    assert b"gfx950" in L.bn_version()


def test_log_chunks_range_kat(ffi):
    """crates/compute/src/layer.rs:827-848: Chunked(256, min 4), ChunkedMut(256, min 6), Local(8) -> 0..2"""
    maps = [("chunked", ffi.DevSlice(0x1000, 256), 4), ("chunked_mut", ffi.DevSlice(0x9000, 256), 6), ("local", 8)]
    r = ffi.log_chunks_range(maps)
    assert (r.start, r.stop) == (0, 2)


def test_host_scalar_field_matches_oracle(ffi, oracle):
    import random

    rng = random.Random(5)
    for _ in range(2000):
        a, b = rng.getrandbits(128), rng.getrandbits(128)
        assert ffi.HostField.mul(a, b) == oracle.mul(a, b)
    # operands confined to sub-fields / single limbs (the table-based Karatsuba of csrc/hostmul.hpp recurses per half)
    for wa in (1, 2, 8, 16, 32, 64, 128):
        for wb in (1, 8, 32, 64, 128):
            for sh in (0, 8, 64, 96):
                a, b = (rng.getrandbits(wa) << sh) & ((1 << 128) - 1), rng.getrandbits(wb)
                assert ffi.HostField.mul(a, b) == oracle.mul(a, b)
                assert ffi.HostField.mul(b, a) == oracle.mul(a, b)
    cases = [1, 2, 3, 0xFF, 1 << 64, (1 << 64) - 1, (1 << 128) - 1] + [1 << i for i in range(0, 128, 7)]
    cases += [rng.getrandbits(w) | 1 for w in (2, 4, 8, 16, 32, 64, 128) for _ in range(20)]
    for a in cases:
        inv = ffi.HostField.invert(a)
        assert ffi.HostField.mul(a, inv) == 1
        assert inv == oracle.invert(a)
    assert ffi.HostField.invert(0) == 0


def test_host_scalar_field_table_route_matches_oracle():
    """The same checks with BN_HOSTMUL=table (csrc/hostmul.hpp's Karatsuba-to-bytes form, the route of hosts without
    PCLMULQDQ); the default route of this host is whatever csrc/hostmul_clmul.cpp's self-check allowed.  A fresh process:
    the route is chosen at the first product."""
    import subprocess
    import sys

    code = (
        "import random, sys\n"
        "sys.path.insert(0, %r)\n"
        "import oracle\n"
        "from binius_amd._ffi import HostField as F\n"
        "rng = random.Random(11)\n"
        "cases = [(0, 5), (1, 1), ((1 << 128) - 1, (1 << 128) - 1)] + [(1 << i, rng.getrandbits(128)) for i in range(128)]\n"
        "cases += [(rng.getrandbits(128), rng.getrandbits(128)) for _ in range(1500)]\n"
        "assert all(F.mul(a, b) == oracle.mul(a, b) for a, b in cases)\n"
        "assert all(F.mul(a, F.invert(a)) == 1 for a, _ in cases[3:200])\n"
        "print('ok')\n"
    ) % ROOT
    for route in ("table", "clmul"):
        env = dict(os.environ, BN_HOSTMUL=route)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.strip() == "ok", (route, out.stderr[-2000:])


def test_twiddle_basis_matches_oracle(ffi, oracle):
    import numpy as np

    for lvl, d in ((3, 8), (4, 10), (4, 16), (5, 24), (5, 32), (6, 40)):
        assert np.array_equal(ffi.ntt_s_evals(lvl, d), oracle.ntt_s_evals(lvl, d))


def test_devslice_and_bump_allocator(ffi):
    """ComputeMemory handle arithmetic (memory.rs:69-234) and BumpAllocator (alloc.rs:123-158)."""
    s = ffi.DevSlice(0x10000, 256)
    lo, hi = s.split_half()
    assert (lo.ptr, lo.len, hi.ptr, hi.len) == (0x10000, 128, 0x10000 + 128 * 16, 128)
    x = s
    while x.len > 1:
        x, _ = x.split_half()
    assert x.len == 1 and x.ptr == 0x10000
    assert [c.ptr for c in s.chunks(64)] == [0x10000 + i * 64 * 16 for i in range(4)]
    bump = ffi.BumpAllocator(s)
    assert bump.alloc(100).len == 100 and bump.alloc(100).len == 100
    with pytest.raises(ffi.BnError) as e:
        bump.alloc(100)
    assert e.value.kind == "Alloc"
    assert bump.capacity() == 56


def test_no_gpu_fails_loudly(ffi):
    """No CPU fallback: without a visible GPU context creation is a DeviceError, never a silent path."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(ffi.BnError) as e:
        ffi.Context(0, 0)
    assert e.value.kind == "DeviceError"


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "binius_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn
                assert "oracle/" not in txt and "_ref.h" not in txt, fn


def test_oracle_is_only_the_checker():
    """Outside tests/, the oracle may be used by __graft_entry__.smoke() (as the checker) and by the
    cpu_baseline leg of bench.py -- nowhere else (tools included)."""
    import re

    for fn in os.listdir(os.path.join(ROOT, "tools")):
        if fn.endswith(".py"):
            txt = open(os.path.join(ROOT, "tools", fn)).read()
            assert not re.search(r"^\s*(import|from)\s+.*\boracle\b", txt, re.M), fn
    bench = open(os.path.join(ROOT, "bench.py")).read()
    imports = [m.start() for m in re.finditer(r"^\s*import oracle\b", bench, re.M)]
    assert len(imports) == 1
    assert bench.index("# ---- CPU baseline") < imports[0], "bench.py imports the oracle outside its cpu_baseline leg"
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert entry.index("def smoke") < entry.index("import oracle")


def test_synthetic_streams_match_the_oracle_generator(oracle):
    from binius_amd import synthetic

    assert np.array_equal(oracle.random_b128(123, 1000), synthetic.random_b128(123, 1000))
    assert oracle.random_scalars(77, 9) == synthetic.random_scalars(77, 9)


def test_host_library_exports_every_declared_symbol(ffi):
    import ctypes

    hdr = open(os.path.join(ROOT, "include", "binius_amd_host.h")).read()
    syms = sorted(set(re.findall(r"\b(bnh_[a-z0-9_]+)\s*\(", hdr)) - {"bnh_round_reduce_fn"})
    assert len(syms) >= 10
    ffi.lib()
    L = ctypes.CDLL(os.path.join(ROOT, "binius_amd", "libbinius_amd_host.so"))
    for s in syms:
        assert hasattr(L, s), "libbinius_amd_host.so does not export %s" % s


def test_design_tables_are_generated_from_the_committed_profiles():
    """DESIGN.md section 5 is generated (tools/gen_design_tables.py) from profiles/r04: hand edits or stale numbers fail here."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_design_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
