#!/usr/bin/env python3
"""End-to-end timing of the device-resident FRI commit phase (commit_interleaved: message repeat + batched additive NTT
+ Groestl Merkle tree, root read back) and of the fold phase (fri_fold + commitment per oracle), at a prover-sized
shape, through the compiled C++ mirror (binius_amd/host/fri.hpp behind bnh_fri_commit_fold).  One JSON line; inputs
from binius_amd.synthetic."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic
from binius_amd._host import FRIParams, FriPlan

ap = argparse.ArgumentParser()
ap.add_argument("--log-dim", type=int, default=20)
ap.add_argument("--log-batch", type=int, default=4)
ap.add_argument("--log-inv-rate", type=int, default=1)
ap.add_argument("--arity", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
log_msg = a.log_dim + a.log_batch
n_arities = (log_msg - 1) // a.arity
while n_arities * a.arity >= log_msg: n_arities -= 1
p = FRIParams(a.log_dim, a.log_inv_rate, a.log_batch, [a.arity] * n_arities, n_test_queries=100)
n_msg = 1 << log_msg
n_code = n_msg << a.log_inv_rate
if __import__("os").environ.get("BN_BIND_NUMA") != "0":
    binius_amd.bind_host_thread_to_device(0)  # (the driving thread on the device's NUMA node)
hal = binius_amd.Context(0, n_msg + 3 * n_code + (1 << 16))
base = hal.dev_alloc()
d_msg = base.alloc(n_msg)
hal.copy_h2d(synthetic.random_b128(0xF21, n_msg), d_msg)
challenges = synthetic.random_scalars(0xC4A, p.n_fold_rounds())
plan = FriPlan(hal, p, d_msg, base.alloc(3 * n_code), challenges)
cs = [plan.run() for _ in range(a.reps + 1)][1:]
commit_ms, fold_ms = min(c[0] for c in cs), min(c[1] for c in cs)
shape = "log_dim %d, log_batch %d, log_inv_rate %d, arities %s" % (a.log_dim, a.log_batch, a.log_inv_rate, p.fold_arities)
print(json.dumps({"op": "FRI commit phase (RS encode + Merkle tree, root read back) and fold phase (%d rounds), " % p.n_fold_rounds() + shape,
                  "commit_ms": round(commit_ms, 3), "fold_ms": round(fold_ms, 3), "codeword_MiB": n_code * 16 >> 20,
                  "commit_codeword_GBps": round(16 * n_code / commit_ms / 1e6, 1)}))
