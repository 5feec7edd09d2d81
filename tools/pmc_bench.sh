#!/bin/bash
# HBM traffic of the bench command from counters-only rocprofv3 passes (FETCH_SIZE and WRITE_SIZE in separate
# passes: they do not fit one TCC pass; no trace domain beside --pmc).  Writes gpurun_out/bench_pmc.json:
# per workload and kernel symbol the average bytes per launch, 2 x FETCH_SIZE (gfx950 correction,
# MI355X_MICROARCH.md section HBM) + WRITE_SIZE, both in KiB.
# usage: tools/pmc_bench.sh <build-id> <n_vars> [<n_vars> ...]
BUILD=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
for NV in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    # (BN_ARM=0: under counter collection every dispatch is serialised; no kernel should sit waiting for the host)
    BN_ARM=0 rocprofv3 --pmc $C --output-format csv -d $OUT/n${NV}_$C -- python $R/bench.py --n-vars $NV --steps 1 --warmup 1 --no-cpu-baseline --no-claim-groups --no-prof > $OUT/n${NV}_$C.log 2>&1
  done
done
python3 - "$OUT" "$BUILD" "$@" <<'PY'
import collections, csv, glob, json, sys
out, build, nvs = sys.argv[1], sys.argv[2], sys.argv[3:]
sys.path.insert(0, out + "/../..")
import bench
res = {"build": build, "csrc_sha16": bench.csrc_sha16(), "units": "bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, averaged over the launches of the kernel symbol in one warm-up + one timed sumcheck (k_foldeval_mfma = both forms of the fused kernel: k_foldeval_mfma_fp4 and k_foldeval_mfma)",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --n-vars N --steps 1 --warmup 1 --no-cpu-baseline --no-prof", "workloads": {}}
for nv in nvs:
    acc = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/n%s_%s/**/*counter_collection.csv" % (out, nv, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != c:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("bn::", "").split("<")[0]
                if k == "k_foldeval_mfma_fp4":  # the two forms of the fused kernel are one class (bench.py: fold_eval_mfma)
                    k = "k_foldeval_mfma"
                if k == "k_roundeval_fp4_ws":  # likewise round 0's two forms (bench.py: round_eval_mfma)
                    k = "k_roundeval_fp4"
                acc[k][c][0] += float(r["Counter_Value"])
                acc[k][c][1] += 1
    w = {}
    for k, v in acc.items():
        if not k.startswith("k_"):
            continue
        nf, nw = v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]
        if nf == 0 or nw == 0:
            continue
        fetch = v["FETCH_SIZE"][0] / nf * 1024 * 2
        write = v["WRITE_SIZE"][0] / nw * 1024
        w[k] = {"launches": nf, "fetch_bytes_per_launch_x2": round(fetch), "write_bytes_per_launch": round(write), "traffic_bytes_per_launch": round(fetch + write)}
    res["workloads"]["n_vars_local=%s,m=2" % nv] = w
json.dump(res, open(out + "/../bench_pmc.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
rm -rf $OUT/n*_FETCH_SIZE $OUT/n*_WRITE_SIZE
