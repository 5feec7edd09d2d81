"""pairwise_product_reduce alone at 2^20 (and smaller trees), under the measurement knobs of abi_ops.cpp."""
import json
import sys
import time

sys.path.insert(0, ".")
import binius_amd  # noqa: E402
from binius_amd import synthetic  # noqa: E402


def main():
    with binius_amd.Context(0, (1 << 22) + 4096) as hal:
        alloc = hal.dev_alloc()
        for log_n in ([int(v) for v in sys.argv[1:]] or [20, 16, 12, 6]):
            n = 1 << log_n
            x = synthetic.random_b128(0x77 + log_n, n)
            dx = alloc.alloc(n)
            hal.copy_h2d(x, dx)
            outs = [alloc.alloc(n >> (r + 1)) for r in range(log_n)]
            for _ in range(5):
                hal.pairwise_product_reduce(dx, outs)
            hal.sync()
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                hal.pairwise_product_reduce(dx, outs)
            hal.sync()
            dt = (time.perf_counter() - t0) / reps
            print(json.dumps({"op": "pairwise_product_reduce", "log_n": log_n, "us": round(dt * 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()
