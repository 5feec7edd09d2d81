#!/usr/bin/env python3
"""Two launches of the MLE-check round evaluation (sum a*b*eq at the points 1 and infinity) at n = 24, for
rocprofv3 --pmc / --kernel-trace runs on k_roundeval9_eq alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd.sumcheck import round_eval_kernel
log_n=24; n=1<<log_n
hal = binius_amd.Context(0, 3*n + (1<<16))
alloc = hal.dev_alloc()
A,B,D = (alloc.alloc(n) for _ in range(3))
for j,s in enumerate((A,B,D)): hal.copy_h2d(synthetic.random_b128(0xB1A50000+j, n), s)
e3 = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1), ("var", 2), ("mul", 2, 3)])
k3, maps3 = round_eval_kernel(log_n, [1], [A, B], [e3], eq_ind=D.slice(0, n//2)); ops3, rets3, lc3 = hal.record(k3, maps3)
for _ in range(2): hal.kernel_launch(maps3, ops3, rets3, lc3)
hal.sync()
