"""GPU parity tests for the additive NTT vs the oracle's restatement of the reference's scalar
NTT (crates/ntt/src/tests/reference.rs:68-160), mirroring crates/ntt/src/tests/ntt_tests.rs:24-186
(agreement with the simple reference, forward/inverse round trip, cosets, batched shapes,
skip_rounds)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = {3: np.uint8, 4: np.uint16, 5: np.uint32, 6: np.uint64}


@pytest.fixture(scope="module")
def hal():
    import binius_amd

    ctx = binius_amd.Context(0, 1 << 21)
    yield ctx
    ctx.close()


def make_data(oracle, seed, n, elem_level):
    if elem_level == 7:
        return oracle.random_b128(seed, n)
    words = oracle.splitmix_words(seed, n)
    return words.astype(DT[elem_level])


@pytest.mark.parametrize(
    "elem_level,tw_level,log_domain,log_x,log_y,log_z,coset,coset_bits,skip",
    [
        (5, 5, 12, 0, 12, 0, 0, 0, 0),
        (5, 5, 14, 0, 10, 0, 5, 3, 0),
        # large single B32 transforms: the bit-sliced path (kernels_ntt_bs.hip), every pass split
        (5, 5, 14, 0, 14, 0, 0, 0, 0),
        (5, 5, 15, 0, 15, 0, 0, 0, 0),
        (5, 5, 19, 0, 16, 0, 5, 3, 0),
        (5, 5, 17, 0, 17, 0, 0, 0, 0),
        (5, 5, 20, 0, 19, 0, 1, 1, 0),
        (5, 5, 20, 0, 20, 0, 0, 0, 0),
        # the same path for batches and for B64 / B128 data (interleaved B32 columns)
        (5, 5, 16, 2, 14, 1, 1, 1, 0),
        (6, 5, 15, 0, 15, 0, 0, 0, 0),
        (7, 5, 16, 0, 15, 0, 1, 1, 0),
        (7, 5, 14, 1, 14, 1, 0, 0, 0),
        # skip_rounds (the RS-encoding shape): fewer in-register layers / fewer lower layers
        (5, 5, 16, 0, 16, 0, 0, 0, 2),
        (5, 5, 16, 0, 15, 0, 1, 1, 7),
        (7, 5, 15, 1, 14, 0, 0, 0, 1),
        (5, 5, 14, 0, 14, 0, 0, 0, 5),
        (5, 5, 12, 2, 8, 1, 1, 2, 0),
        (5, 5, 12, 0, 10, 0, 0, 0, 3),
        (7, 5, 12, 0, 10, 0, 0, 1, 0),
        (7, 5, 13, 2, 9, 0, 3, 2, 1),
        (4, 4, 10, 0, 10, 0, 0, 0, 0),
        (6, 5, 10, 1, 8, 1, 1, 1, 0),
        (7, 4, 9, 0, 9, 0, 0, 0, 0),
    ],
)
def test_ntt_forward_inverse(hal, oracle, elem_level, tw_level, log_domain, log_x, log_y, log_z, coset, coset_bits, skip):
    import binius_amd

    n = 1 << (log_x + log_y + log_z)
    data = make_data(oracle, 0x0177, n, elem_level)
    s = binius_amd.ntt_s_evals(tw_level, log_domain)
    assert np.array_equal(s, oracle.ntt_s_evals(tw_level, log_domain))
    alloc = hal.dev_alloc()
    nbytes = data.nbytes
    d = alloc.alloc((nbytes + 15) // 16)
    hal.copy_bytes_h2d(data, d.ptr)
    hal.ntt_forward(d.ptr, elem_level, tw_level, s, log_domain, log_x, log_y, log_z, coset, coset_bits, skip)
    got = hal.copy_bytes_d2h(d.ptr, np.zeros_like(data))
    exp = data.copy()
    assert oracle.ntt_forward(exp, elem_level, tw_level, s, log_domain, log_x, log_y, log_z, coset, coset_bits, skip) == 0
    assert np.array_equal(got, exp)
    hal.ntt_inverse(d.ptr, elem_level, tw_level, s, log_domain, log_x, log_y, log_z, coset, coset_bits, skip)
    back = hal.copy_bytes_d2h(d.ptr, np.zeros_like(data))
    assert np.array_equal(back, data)


def test_ntt_validation(hal):
    import binius_amd

    s = binius_amd.ntt_s_evals(5, 10)
    alloc = hal.dev_alloc()
    d = alloc.alloc(1 << 10)
    with pytest.raises(binius_amd.BnError):
        hal.ntt_forward(d.ptr, 5, 5, s, 10, 0, 11, 0)  # domain too small
    with pytest.raises(binius_amd.BnError):
        hal.ntt_forward(d.ptr, 5, 5, s, 10, 0, 8, 0, coset=4, coset_bits=2)  # coset out of bounds
    with pytest.raises(binius_amd.BnError):
        hal.ntt_forward(d.ptr, 4, 5, s, 10, 0, 8, 0)  # twiddle field larger than element field
