#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command; leaves kernel_stats.csv + per_launch.jsonl (last 200 launches) in gpurun_out/<name>
# usage: tools/trace_cmd.sh <name> <command...>
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$NAME
rm -rf $OUT; mkdir -p $OUT
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- "$@" > $OUT/stdout.log 2> $OUT/stderr.log)
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
for f in glob.glob(out + "/raw/**/*kernel_stats.csv", recursive=True):
    open(out + "/kernel_stats.csv", "w").write(open(f).read())
rows = []
for f in glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
with open(out + "/per_launch.jsonl", "w") as g:
    for s, d, k, grid in rows[-200:]:
        g.write(json.dumps({"kernel": k, "us": round(d / 1000, 2), "grid": grid}) + "\n")
PY
rm -rf $OUT/raw
