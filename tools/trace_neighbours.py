#!/usr/bin/env python3
"""Which kernels run right before / after every launch of a given kernel (default __amd_rocclr_copyBuffer) in a rocprofv3
--kernel-trace CSV: finds the host-side call that issues unexplained copies / fills.  Usage: trace_neighbours.py <dir> [name]"""
import collections
import csv
import glob
import sys

name = sys.argv[2] if len(sys.argv) > 2 else "__amd_rocclr_copyBuffer"
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]  # noqa: E731
before, after, sizes = collections.Counter(), collections.Counter(), collections.Counter()
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[r.get("Stream_Id", r.get("Queue_Id", "0"))].append(r)
for s, rs in by_stream.items():
    for i, r in enumerate(rs):
        if name in r["Kernel_Name"]:
            before[short(rs[i - 1]["Kernel_Name"]) if i else "-"] += 1
            after[short(rs[i + 1]["Kernel_Name"]) if i + 1 < len(rs) else "-"] += 1
            sizes[(r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))] += 1
print("launches of", name, sum(before.values()))
print("preceded by:", before.most_common(12))
print("followed by:", after.most_common(12))
print("grid sizes:", sizes.most_common(8))
