"""Counters of the LONGEST dispatch of a kernel in rocprofv3 --pmc output (the pruned full scan among its pilot launches).
usage: pmc_pick.py DIR SUBSTR -> JSON {counter: value, "_duration_ns": ...} per DIR/*/ run, merged"""
import csv
import glob
import json
import sys

d, sub = sys.argv[1], sys.argv[2]
out = {}
for cc in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(cc)) if sub in r["Kernel_Name"]]
    if not rows:
        continue
    dur = {}
    for r in rows:
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if "End_Timestamp" in r else 0
    if not any(dur.values()):
        kt = glob.glob(cc.replace("counter_collection", "kernel_trace"))
        if kt:
            for r in csv.DictReader(open(kt[0])):
                if r.get("Dispatch_Id") in dur:
                    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    best = max(dur, key=dur.get)
    for r in rows:
        if r["Dispatch_Id"] == best:
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out.setdefault("_duration_ns", []).append(dur[best])
print(json.dumps(out, indent=1))
