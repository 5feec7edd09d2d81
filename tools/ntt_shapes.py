#!/usr/bin/env python3
"""Timing of the bit-sliced additive NTT across element widths and interleavings (B32 / B64 / B128 data,
batched columns, the Reed-Solomon encoding shape).  One JSON line per shape; inputs from binius_amd.synthetic."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
hal = binius_amd.Context(0, (1 << 25) + (1 << 12))
alloc = hal.dev_alloc()
D = alloc.alloc(1 << 25)
hal.copy_h2d(synthetic.random_b128(1, 1 << 24), D.slice(0, 1 << 24))
def timed(name, fn):
    ts = []
    for _ in range(4):
        hal.sync(); hal.timer_begin(); fn(); ts.append(hal.timer_end_ms())
    print(json.dumps({"op": name, "ms": round(min(ts[1:]), 4)}), flush=True)
s = binius_amd.ntt_s_evals(5, 24)
timed("B32 2^24 lx0", lambda: hal.ntt_forward(D.ptr, 5, 5, s, 24, 0, 24, 0))
timed("B64 2^23 (lx1)", lambda: hal.ntt_forward(D.ptr, 6, 5, s, 24, 0, 23, 0))
timed("B128 2^22 (lx2)", lambda: hal.ntt_forward(D.ptr, 7, 5, s, 24, 0, 22, 0))
timed("B128 2^18 x 2^4 batch (lx6)", lambda: hal.ntt_forward(D.ptr, 7, 5, s, 24, 4, 18, 0))
s21 = binius_amd.ntt_s_evals(5, 21)
timed("RS encode shape: 2^25 B128, log_x 6, log_y 21, skip 1", lambda: hal.ntt_forward(D.ptr, 5, 5, s21, 21, 6, 21, 0, 0, 0, 1))
