#!/usr/bin/env python3
"""Per-op timings of every ComputeLayer entry point at a representative size, with the
algorithmic-bytes roofline fraction (SURVEY.md section 8d figures).  Output: one JSON line per op."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import binius_amd
from binius_amd import synthetic
from binius_amd.sumcheck import bivariate_product_expr, round_eval_kernel

ap = argparse.ArgumentParser()
ap.add_argument("--log-n", type=int, default=24)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
n = 1 << a.log_n
if __import__("os").environ.get("BN_BIND_NUMA") != "0":
    binius_amd.bind_host_thread_to_device(0)  # (the driving thread on the device's NUMA node)
hal = binius_amd.Context(0, 5 * n + (1 << 16))
alloc = hal.dev_alloc()
A, B, Cc, D = (alloc.alloc(n) for _ in range(4))
for j, s in enumerate((A, B)):
    hal.copy_h2d(synthetic.random_b128(0xB1A50000 + j, n), s)
hal.copy_d2d(A, Cc); hal.copy_d2d(B, D)
z = synthetic.random_scalars(0xC4A1, 1)[0]

def timed(name, fn, alg_bytes, setup=None, note=""):
    ts = []
    for _ in range(a.reps + 1):
        if setup: setup()
        hal.sync(); hal.timer_begin(); fn(); ts.append(hal.timer_end_ms())
    ms = min(ts[1:])
    gbs = alg_bytes / ms / 1e6
    print(json.dumps({"op": name, "ms": round(ms, 4), "alg_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4), "note": note}), flush=True)

half = n // 2
timed("extrapolate_line (fold), 2^%d outputs" % (a.log_n - 1), lambda: hal.extrapolate_line(Cc.slice(0, half), Cc.slice(half, n), z), 24 * n)
expr = bivariate_product_expr(hal, 0, 1)
k, maps = round_eval_kernel(a.log_n, [1], [A, B], [expr]); ops, rets, lc = hal.record(k, maps)
timed("accumulate_kernels round-eval a*b, m=2, 2^%d" % a.log_n, lambda: hal.kernel_launch(maps, ops, rets, lc), 16 * 2 * n)
pt = synthetic.random_scalars(7, a.log_n - 1)
def te_setup(): hal.fill(Cc.slice(0, 1), 1)
timed("tensor_expand 0 -> %d vars" % (a.log_n - 1), lambda: hal.tensor_expand(0, pt, Cc.slice(0, half)), 3 * 16 * half, te_setup)
e3 = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)])
k3, maps3 = round_eval_kernel(a.log_n, [1], [A, B], [e3], eq_ind=D.slice(0, half)); ops3, rets3, lc3 = hal.record(k3, maps3)
timed("accumulate_kernels MLE-check a*b*eq, 2^%d" % a.log_n, lambda: hal.kernel_launch(maps3, ops3, rets3, lc3), 16 * 2 * n + 16 * half)
timed("inner_product F x F, 2^%d" % a.log_n, lambda: hal.inner_product(A, 7, B), 32 * n)
timed("inner_product B32 x F, 2^%d" % a.log_n, lambda: hal.inner_product(A.slice(0, n // 4), 5, B), 16 * n + 4 * n)
timed("inner_product B1 x F, 2^%d" % a.log_n, lambda: hal.inner_product(A.slice(0, n // 128), 0, B), 16 * n + n // 8)
def am(ke, lc_, b): ke.add_assign(a.log_n - 1 - lc_, b[1].to_ref(), b[0])
mm = [("chunked_mut", Cc.slice(0, half), 0), ("chunked", Cc.slice(half, n), 0)]
timed("map_kernels add_assign, 2^%d" % (a.log_n - 1), lambda: hal.map_kernels(am, mm), 48 * half)
vec = alloc.alloc(64); hal.copy_h2d(synthetic.random_b128(9, 64), vec)
out = alloc.alloc(n // 64 * 4)
timed("fold_right B32 matrix 2^%d x vec 2^6" % (a.log_n + 2 - 6), lambda: hal.fold_right(A, 5, vec, out), 16 * n + 16 * (n // 16))
timed("fold_left  B32 matrix, vec 2^6", lambda: hal.fold_left(A, 5, vec, out), 16 * n + 16 * (n // 16))
s5 = binius_amd.ntt_s_evals(5, 28)
lb = 4
ch = synthetic.random_scalars(11, lb + 3)
fo = alloc.alloc(n >> (lb + 3))
timed("fri_fold log_len=%d log_batch=4, 3 fold rounds" % (a.log_n - lb), lambda: hal.fri_fold(s5, 5, 28, a.log_n - lb, lb, ch, A, fo), 16 * n + 16 * (n >> 7))
small = 1 << 20
timed("compute_composite a*b, 2^20", lambda: hal.compute_composite([A.slice(0, small), B.slice(0, small)], Cc.slice(0, small), expr), 48 * small)
timed("compute_composite a*b, 2^%d" % (a.log_n - 1), lambda: hal.compute_composite([A.slice(0, half), B.slice(0, half)], Cc.slice(0, half), expr), 48 * half)
# generic circuits (csrc/abi_circuit.cpp: compiled into passes of the throughput kernels)
gen = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3), ("add", 4, 0)])  # a*b*c + a
q = n // 4
timed("compute_composite a*b*c + a (generic circuit), 2^%d" % (a.log_n - 2), lambda: hal.compute_composite([A.slice(0, q), B.slice(0, q), D.slice(0, q)], Cc.slice(0, q), gen), 64 * q)
def gk(ke, lc_, b):
    acc = ke.decl_value(0)
    ke.sum_composition_evals([x.to_ref() for x in b], gen, 1, acc)
    return [acc]
gm = [("chunked", A.slice(0, q), 0), ("chunked", B.slice(0, q), 0), ("chunked", D.slice(0, q), 0)]
gops, grets, glc = hal.record(gk, gm)
timed("accumulate_kernels sum of a*b*c + a (generic circuit), 2^%d rows" % (a.log_n - 2), lambda: hal.kernel_launch(gm, gops, grets, glc), 48 * q)
outs = [alloc.alloc(small >> (r + 1)) for r in range(20)] if alloc.capacity() > small else None
if outs:
    timed("pairwise_product_reduce 2^20", lambda: hal.pairwise_product_reduce(A.slice(0, small), outs), 16 * small * 2)
s24 = binius_amd.ntt_s_evals(5, a.log_n)
timed("NTT forward 2^%d x B32" % a.log_n, lambda: hal.ntt_forward(Cc.ptr, 5, 5, s24, a.log_n, 0, a.log_n, 0), 2 * 4 * n)
timed("NTT forward 2^%d x B128 (B32 twiddles)" % (a.log_n - 2), lambda: hal.ntt_forward(Cc.ptr, 7, 5, s24, a.log_n, 0, a.log_n - 2, 0), 2 * 16 * (n // 4))
hal.close()
