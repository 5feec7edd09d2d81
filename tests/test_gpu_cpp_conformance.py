"""Runs the C++ ports of the reference's backend-conformance tests (tests/cpp/conformance.cpp:
crates/compute_test_utils/src/layer.rs + bivariate_sumcheck.rs against the C++ host mirror
binius_amd/host/compute_layer.hpp).  The binary is built by __graft_entry__.build()."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "conformance")


@pytest.mark.gpu
def test_cpp_conformance_suite():
    assert os.path.exists(BIN), "tests/cpp/conformance missing -- run __graft_entry__.build()"
    p = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "20/20 conformance tests passed" in p.stdout


def test_cpp_conformance_binary_is_built():
    import __graft_entry__ as g

    g.build()
    assert os.path.exists(BIN)
