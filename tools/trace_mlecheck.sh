#!/bin/bash
# kernel timeline of one MLE-check prove (n = 24): rocprofv3 kernel trace -> per-kernel start/duration/gap
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mle_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/bench_mlecheck.py --n-vars ${1:-24} --steps 1 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last run = last occurrence of the first kernel name pattern (k_mul9 launches mark a run start)
idx = [i for i, r in enumerate(rows) if "k_mul9" in r["Kernel_Name"]]
start = idx[-2] if len(idx) >= 2 else 0
prev_end = None
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:60]))
    prev_end = e
PY
