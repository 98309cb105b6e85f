#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name (last dispatch of each)."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
for f in files:
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        key = (r["Kernel_Name"][:60], r["Dispatch_Id"])
        d = agg.setdefault(key, {"_dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "_grid": r["Grid_Size"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    last = {}
    for (name, did), d in agg.items():
        last[name] = d
    for name, d in last.items():
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        print(name, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()})
