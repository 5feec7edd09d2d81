#!/usr/bin/env python3
"""fri_fold at the benchmarked shape (log_len 20 x log_batch 4, 3 fold rounds) a few times -- a target for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

n = 1 << 24
hal = binius_amd.Context(0, 3 * n)
alloc = hal.dev_alloc()
A = alloc.alloc(n)
hal.copy_h2d(synthetic.random_b128(1, n), A)
lb = 4
s5 = binius_amd.ntt_s_evals(5, 28)
ch = synthetic.random_scalars(11, lb + 3)
fo = alloc.alloc(n >> (lb + 3))
for _ in range(4):
    hal.fri_fold(s5, 5, 28, 24 - lb, lb, ch, A, fo)
hal.sync()
hal.close()
