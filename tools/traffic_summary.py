#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md prescribes for gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE reports
half of a wide coalesced read stream; WRITE_SIZE is exact for 16-B/lane stores and float atomics).
Usage: traffic_summary.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


F, W = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in F:
    if k not in W:
        continue
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    f, w = sum(F[k]) / len(F[k]), sum(W[k]) / len(W[k])
    out[short] = {"launches": len(F[k]), "fetch_size_kb_avg": round(f, 1), "write_size_kb_avg": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(k, v)
