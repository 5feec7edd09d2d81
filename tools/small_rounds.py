"""Per-round latency of small sumchecks (everything is launch/latency bound): wall time per round of
the compiled host prover at n_vars = 4..12.  Usage: python tools/small_rounds.py"""
import json
import sys
import time

sys.path.insert(0, ".")
import binius_amd  # noqa: E402
from binius_amd import synthetic
from binius_amd._host import SumcheckPlan  # noqa: E402


def main():
    import os

    if os.environ.get("BN_BIND_NUMA") != "0":
        print("host thread:", binius_amd.bind_host_thread_to_device(0), file=sys.stderr)
    hal = binius_amd.Context(0, 1 << 16)
    for n_vars in (4, 8, 12):
        alloc = hal.dev_alloc()
        mls = [synthetic.random_b128(0xB1A50000 + j, 1 << n_vars) for j in range(2)]
        d = []
        for x in mls:
            s = alloc.alloc(1 << n_vars)
            hal.copy_h2d(x, s)
            d.append(s)
        scratch = alloc.alloc(1 << n_vars)
        stream = synthetic.random_scalars(0xC4A1, n_vars + 1)
        plan = SumcheckPlan(hal, n_vars, d, scratch, [(0, 1)], [0], stream[0], stream[1:])
        for _ in range(20):
            plan.run()
        hal.sync()
        iters = 300
        t0 = time.perf_counter()
        for _ in range(iters):
            plan.run()
        hal.sync()
        dt = time.perf_counter() - t0
        c = hal.arm_counters()
        rec = {"n_vars": n_vars, "us_per_sumcheck": round(dt / iters * 1e6, 2), "us_per_round": round(dt / iters / n_vars * 1e6, 2)}
        if c["hits"]:  # armed rounds (csrc/arm.hpp): host time from handing over z to seeing the result, per armed round
            rec["armed_rounds"] = c["hits"]
            rec["us_go_to_result"] = round(c["ns_wait"] / c["hits"] / 1e3, 2)
            rec["us_of_which_enqueue_next"] = round(c["ns_launch"] / c["hits"] / 1e3, 2)
            rec["us_launch_entry_to_go"] = round(c["ns_parse"] / c["hits"] / 1e3, 2)
        print(json.dumps(rec))
    hal.close()


if __name__ == "__main__":
    main()
