"""Sum rocprofv3 --pmc counter_collection CSVs per counter for kernels whose name contains a substring.
usage: pmc_sum.py DIR SUBSTR  -> JSON {counter: mean value per launch}"""
import csv
import glob
import json
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
tot, cnt = defaultdict(float), defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]].add((f, r["Dispatch_Id"]))
print(json.dumps({k: tot[k] / max(len(cnt[k]), 1) for k in sorted(tot)}, indent=1))
