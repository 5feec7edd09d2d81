#!/usr/bin/env python3
"""Time the forward additive NTT (config 3: 2^24 BinaryField32b, shape {0,24,0}) and others."""
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
hal.close()
