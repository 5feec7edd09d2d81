#!/bin/bash
# rocprofv3 kernel trace of one bench.py command; leaves kernel_stats + per-launch durations in gpurun_out/<name>
# usage: tools/trace_bench.sh <name> <bench args...>
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python $R/bench.py "$@" > $OUT/bench_line.json 2> $OUT/stderr.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
for f in glob.glob(out + "/raw/**/*kernel_stats.csv", recursive=True):
    open(out + "/kernel_stats.csv", "w").write(open(f).read())
rows = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
with open(out + "/per_launch.jsonl", "w") as g:
    for s, d, k, grid in rows[-120:]:
        g.write(json.dumps({"kernel": k, "us": round(d / 1000, 2), "grid": grid}) + "\n")
PY
rm -rf $OUT/raw
tail -c 1500 $OUT/bench_line.json
