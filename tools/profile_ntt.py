#!/usr/bin/env python3
"""Time the forward additive NTT (config 3: 2^24 BinaryField32b, shape {0,24,0}) and others.

The transform is VALU-bound, so besides the HBM figure the last line is a JSON roofline block with
"bound": "valu": algorithmic lane-operations = (L - 5) bit-sliced layers x 2^(L-6) plane-set butterflies x 1014
lane-operations each (instruction count of the compiled butterfly loop of k_ntt_bs_pass: the 880-instruction product of
bitslice.hpp -- 1015 until the GF(4)/GF(16) levels were written out on the 3-input LUT --, 64 XORs of the butterfly,
~210 of twiddle construction, bit-field extracts and LDS addressing; one lane-operation processes 32 elements) plus the
word-level head (5 layers x 2^(L-1) butterflies x ~30 lane-operations), against the measured integer VALU peak of
60 T lane-operations/s (profiles/r01/valu_issue_rate.txt)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import binius_amd
from binius_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=24)
ap.add_argument("--elem-level", type=int, default=5)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
n = 1 << a.log_n
nbytes = n * (1 << (a.elem_level - 3))
hal = binius_amd.Context(0, nbytes // 16 + 4096)
alloc = hal.dev_alloc()
d = alloc.alloc(nbytes // 16)
s = binius_amd.ntt_s_evals(5, a.log_n if a.log_n <= 32 else 32)
if a.elem_level == 7:
    data = synthetic.random_b128(0x0177, n)
else:
    data = synthetic.splitmix_words(0x0177, n).astype({5: np.uint32, 6: np.uint64}[a.elem_level])
hal.copy_bytes_h2d(data, d.ptr)
hal.ntt_forward(d.ptr, a.elem_level, 5, s, a.log_n, 0, a.log_n, 0)
hal.sync()
if os.environ.get("BN_NTT_AB"):
    # A/B of the barrier forms on this box, alternating (BN_NTT_WG_BARRIERS is read per call)
    for rep in range(6):
        for mode in ("1", "0"):
            os.environ["BN_NTT_WG_BARRIERS"] = mode
            hal.copy_bytes_h2d(data, d.ptr)
            hal.sync()
            hal.prof_begin()
            hal.ntt_forward(d.ptr, a.elem_level, 5, s, a.log_n, 0, a.log_n, 0)
            print("BN_NTT_WG_BARRIERS=%s: %.4f ms" % (mode, hal.prof_end()["ntt"][0]))
    os.environ.pop("BN_NTT_WG_BARRIERS", None)
for _ in range(a.reps):
    hal.copy_bytes_h2d(data, d.ptr)
    hal.sync()
    t0 = time.perf_counter()
    hal.prof_begin()
    hal.ntt_forward(d.ptr, a.elem_level, 5, s, a.log_n, 0, a.log_n, 0)
    p = hal.prof_end()
    ms = p["ntt"][0]
    print("forward NTT 2^%d x %d-bit: %.3f ms  (%.1f GB/s of the 2*w*2^L algorithmic bytes, %.2f%% of 8 TB/s)" % (
        a.log_n, 1 << a.elem_level, ms, 2 * nbytes / ms / 1e6, 2 * nbytes / ms / 1e6 / 80))
import json
L = a.log_n
cols = 1 << (a.elem_level - 5)  # a larger field is 2 or 4 interleaved B32 columns
lane_ops = cols * ((L - 5) * (1 << (L - 6)) * 1014 + 5 * (1 << (L - 1)) * 30)
print(json.dumps({"op": "forward NTT 2^%d x B%d" % (L, 1 << a.elem_level), "ms": round(ms, 4),
                  "roofline": {"bound": "valu", "achieved": round(lane_ops / ms / 1e9, 2), "peak": 60.0, "unit": "T lane-op/s",
                               "frac": round(lane_ops / ms / 1e9 / 60.0, 4), "lane_ops": lane_ops},
                  "hbm": {"algorithmic_bytes": 2 * nbytes, "GBps": round(2 * nbytes / ms / 1e6, 1)}}))
hal.close()
