#!/bin/bash
# HBM traffic of the group kernel's launches, one by one, from counters-only rocprofv3 passes (FETCH_SIZE and WRITE_SIZE in
# separate passes; no trace domain beside --pmc) over tools/bench_piop.py claims.  Writes gpurun_out/<name>.json: per launch of
# k_group_fp4 (in dispatch order, the last prove of the run) 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md section HBM)
# + WRITE_SIZE in bytes beside the algorithmic bytes of that launch.
# usage: tools/pmc_claims.sh <name> <n_vars> <k> <kind> [env assignments for the command...]
NAME=$1; NV=$2; K=$3; KIND=$4; shift 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$NAME
rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  env "$@" rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python $R/tools/bench_piop.py claims --n-vars $NV --k $K --kind $KIND --group 1 --steps 1 --warmup 0 > $OUT/$C.log 2>&1
done
python3 - "$OUT" "$NAME" "$NV" "$K" "$KIND" "$*" <<'PY'
import csv, glob, json, sys
out, name, nv, k, kind, envs = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, c), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "k_group_fp4" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    per[c] = [v for _, v in rows]
n = min(len(per["FETCH_SIZE"]), len(per["WRITE_SIZE"]))
launches = [round((2 * per["FETCH_SIZE"][i] + per["WRITE_SIZE"][i]) * 1024) for i in range(n)]
# the command proves twice (one timed step, one step under the context's profiler): the second half of the list is the last prove
one = n // 2 if n % 2 == 0 else n
last = launches[n - one:]
m = {"disjoint": 2 * k, "piop": 2 * k, "bipartite": 2 * int(round(k ** 0.5))}[kind]
N = 1 << nv
res = {"name": name, "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python tools/bench_piop.py claims --n-vars %d --k %d --kind %s --group 1 --steps 1 --warmup 0 (%s)" % (nv, k, kind, envs or "default switches"),
       "units": "bytes per launch of k_group_fp4 = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, launches of the last of the run's two proves in dispatch order",
       "launches_seen": n, "m": m, "traffic_bytes": last}
if kind == "disjoint":
    # launch 0: k evaluate jobs over all m arrays (16 * m * N); launch i >= 1: k fold + evaluate jobs on arrays of N / 2^(i-1) elements (24 * m * N / 2^(i-1))
    alg = [16 * m * N] + [24 * m * (N >> (i - 1)) for i in range(1, len(last))]
    res["algorithmic_bytes"] = alg
    res["traffic_over_algorithmic"] = [round(t / a, 4) for t, a in zip(last, alg)]
json.dump(res, open(out + "/../pmc_%s.json" % name, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
