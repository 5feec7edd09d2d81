import sys, json, time
sys.path.insert(0, ".")
import numpy as np
import binius_amd, oracle
from binius_amd import synthetic
from binius_amd._host import SumcheckPlan
hal = binius_amd.Context(0, 1 << 22)
for n_vars in (3, 4, 7, 12, 16, 19):
    alloc = hal.dev_alloc()
    mls = [synthetic.random_b128(0xB1A50000 + j, 1 << n_vars) for j in range(2)]
    d = []
    for x in mls:
        s = alloc.alloc(1 << n_vars); hal.copy_h2d(x, s); d.append(s)
    scratch = alloc.alloc(1 << n_vars)
    stream = synthetic.random_scalars(0xC4A1, n_vars + 1)
    plan = SumcheckPlan(hal, n_vars, d, scratch, [(0, 1)], [0], stream[0], stream[1:])
    c0 = hal.arm_counters()
    r1 = plan.run()
    c1 = hal.arm_counters()
    r2 = plan.run()
    print(n_vars, {k: c1[k]-c0[k] for k in c0})
hal.close()
