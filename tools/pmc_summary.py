#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output directories: per kernel and counter, launches and the average
counter value per launch.  Usage: python tools/pmc_summary.py out.json dir1 [dir2 ...]"""
import collections, csv, glob, json, sys

out = collections.defaultdict(dict)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
        for (kn, cn), (n, v) in acc.items():
            out[kn][cn] = {"launches": n, "avg_per_launch": round(v / n, 1)}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
