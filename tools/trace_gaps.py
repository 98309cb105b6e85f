#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel-trace CSV (one stream): busy vs span over the last `frac` of the
trace, the gap histogram, and the gaps charged to the kernel that FOLLOWS them.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [frac=0.4]"""
import csv
import sys
from collections import defaultdict


def main():
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
    rows = rows[int(len(rows) * (1 - frac)):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms ({100 * (span - busy) / span:.1f} %)")
    hist = defaultdict(lambda: [0, 0])
    after = defaultdict(lambda: [0, 0])
    prev_end = rows[0][1]
    for s, e, n in rows[1:]:
        gap = max(0, s - prev_end)
        prev_end = max(prev_end, e)
        b = 0 if gap < 1000 else 1 if gap < 2000 else 2 if gap < 4000 else 3 if gap < 8000 else 4 if gap < 20000 else 5
        hist[b][0] += 1
        hist[b][1] += gap
        k = n.replace("(anonymous namespace)::", "").replace("void ", "")[:50]
        after[k][0] += 1
        after[k][1] += gap
    names = ["<1us", "1-2us", "2-4us", "4-8us", "8-20us", ">20us"]
    for b in sorted(hist):
        print(f"  gap {names[b]:7s} n {hist[b][0]:6d}  total {hist[b][1] / 1e6:7.2f} ms")
    for k, (n, g) in sorted(after.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  before {k:50s} n {n:5d} idle {g / 1e6:7.2f} ms  avg {g / n / 1e3:6.2f} us")


if __name__ == "__main__":
    main()
