#!/usr/bin/env python3
"""fold_right / fold_left at the benchmark shape, a few calls each -- for kernel traces (tools/trace_cmd.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

n = 1 << 24
hal = binius_amd.Context(0, n + n // 16 + 4096)
alloc = hal.dev_alloc()
A = alloc.alloc(n)
step = 1 << 22
for off in range(0, n, step):
    hal.copy_h2d(synthetic.random_b128_shard(0xF01D, step, 1, 0, start=off), A.slice(off, off + step))
vec = alloc.alloc(64)
hal.copy_h2d(synthetic.random_b128(9, 64), vec)
out = alloc.alloc(n // 64 * 4)
for _ in range(6):
    hal.fold_right(A, 5, vec, out)
hal.sync()
for _ in range(3):
    hal.fold_left(A, 5, vec, out)
hal.sync()
hal.close()
