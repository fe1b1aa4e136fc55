#!/usr/bin/env python3
"""Average per-dispatch PMC values of the k_append* / k_spr* kernels from rocprofv3 counter_collection CSVs."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_append" in k or "k_spr" in k:
            agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
try:
    b = json.load(open(out + "/bench.json"))
    print("bench kernel_ms", b["roofline"]["kernel_ms"], "value", b["value"])
except Exception as e:
    print("no bench.json", e)
for k, d in agg.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print(f"   {c:24s} {sum(v)/len(v):16.1f}  (n={len(v)})")
