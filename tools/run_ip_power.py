#!/usr/bin/env python3
"""inner_product of two 2^27-element vectors (the FP4 round-evaluation kernel alone, 4 GiB read per call) on all-zero, sparse and
random inputs: does the kernel's speed depend on the data (a power-limited clock) or only on the instruction stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 27
n = 1 << lg
hal = binius_amd.Context(0, 2 * n + 4096)
alloc = hal.dev_alloc()
A = alloc.alloc(n)
B = alloc.alloc(n)
step = 1 << 22

def fill_random(S, seed):
    for off in range(0, n, step):
        hal.copy_h2d(synthetic.random_b128_shard(seed, step, 1, 0, start=off), S.slice(off, off + step))

def run(name):
    for _ in range(3):
        hal.inner_product(A, 7, B)
    hal.sync()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        r = hal.inner_product(A, 7, B)
    hal.sync()
    dt = (time.perf_counter() - t0) / reps
    print("%-28s %.3f ms  %.1f GB/s" % (name, dt * 1e3, 32.0 * n / dt / 1e9), flush=True)

hal.fill(A, 0)
hal.fill(B, 0)
run("zeros")
hal.fill(A, 1)
hal.fill(B, 1)
run("ones (1 bit per element)")
fill_random(A, 0xA)
fill_random(B, 0xB)
run("random")
hal.fill(A, 0)
run("a = 0, b random")
hal.close()
