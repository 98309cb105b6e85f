#!/usr/bin/env python3
"""Group a rocprofv3 kernel-trace CSV by (kernel, grid, workgroup): calls, total and average time.
Usage: python tools/kernel_trace_groups.py <kernel_trace.csv> [name-substring] [top]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    g = defaultdict(lambda: [0, 0.0])
    total = 0.0
    for r in csv.DictReader(open(path)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        total += d
        if sub and sub not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        k = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("LDS_Block_Size", ""))
        g[k][0] += 1
        g[k][1] += d
    print(f"all kernels: {total / 1e3:.1f} ms")
    for k, (n, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:60s} grid {k[1]:>8s}x{k[2]}x{k[3]} wg {k[4]:>4s} lds {k[5]:>6s}  calls {n:5d}  total {t / 1e3:8.2f} ms  avg {t / n:9.1f} us")


if __name__ == "__main__":
    main()
