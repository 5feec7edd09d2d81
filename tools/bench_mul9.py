#!/usr/bin/env python3
"""The element-wise product kernel by size (compute_composite a*b, crates/compute/src/layer.rs:552-593): one JSON line per size.
A measurement build (make BN_KNOBS=1) reads BN_MUL9_WPS=3 (three waves per SIMD) and BN_MUL9_DUAL=0 (one wave-batch per rebuild)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import binius_amd
from binius_amd import synthetic

n = 1 << 23
hal = binius_amd.Context(0, 3 * n + 4096)
alloc = hal.dev_alloc()
A, B, C = alloc.alloc(n), alloc.alloc(n), alloc.alloc(n)
for buf, seed in ((A, 1), (B, 2)):
    for off in range(0, n, 1 << 22):
        hal.copy_h2d(synthetic.random_b128_shard(0xE0 + seed, 1 << 22, 1, 0, start=off), buf.slice(off, off + (1 << 22)))
expr = hal.compile_expr([("var", 0), ("var", 1), ("mul", 0, 1)])
for lg in (14, 16, 17, 18, 19, 20, 21, 23):
    m = 1 << lg
    ts = []
    for _ in range(8):
        hal.sync(); hal.timer_begin(); hal.compute_composite([A.slice(0, m), B.slice(0, m)], C.slice(0, m), expr); ts.append(hal.timer_end_ms())
    ms = min(ts[1:])
    print(json.dumps({"log_n": lg, "us": round(ms * 1e3, 2), "G_products_per_s": round(m / ms / 1e6, 2), "wps": os.environ.get("BN_MUL9_WPS", "2"), "dual": os.environ.get("BN_MUL9_DUAL", "1")}), flush=True)
hal.close()
