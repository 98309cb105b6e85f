#!/usr/bin/env python3
"""Where a multi-stream step leaves the chip under-filled: sweep a rocprofv3 kernel-trace CSV and split the span of its last `frac` into
  idle (no kernel running) / thin (every running kernel has fewer workgroups than the chip has CUs) / full (the rest),
and list, for the thin time, which kernels were running (a kernel is charged the thin time it was part of).
Usage: python tools/trace_concurrency.py <kernel_trace.csv> [frac=0.5] [cus=256]"""
import csv
import sys
from collections import defaultdict


def main():
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    cus = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) * max(1, int(r.get("Grid_Size_Y") or 1)) * max(1, int(r.get("Grid_Size_Z") or 1))
        wx = max(1, int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)) * max(1, int(r.get("Workgroup_Size_Y") or 1)) * max(1, int(r.get("Workgroup_Size_Z") or 1))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], gx // wx))
    rows.sort()
    rows = rows[int(len(rows) * (1 - frac)):]
    ev = []
    for i, (s, e, n, w) in enumerate(rows):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    running = set()
    t_prev = ev[0][0]
    idle = thin = full = 0
    thin_by = defaultdict(int)
    conc_hist = defaultdict(int)
    for t, kind, i in ev:
        dt = t - t_prev
        if dt > 0:
            if not running:
                idle += dt
            elif all(rows[k][3] < cus for k in running):
                thin += dt
                for k in running:
                    thin_by[rows[k][2]] += dt
            else:
                full += dt
            conc_hist[min(len(running), 6)] += dt
        t_prev = t
        if kind:
            running.add(i)
        else:
            running.discard(i)
    span = idle + thin + full
    print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms: idle {idle / 1e6:.2f} ms ({100 * idle / span:.1f} %)  thin {thin / 1e6:.2f} ms ({100 * thin / span:.1f} %)  full {full / 1e6:.2f} ms")
    print("  concurrency (kernels running at once): " + "  ".join(f"{k}{'+' if k == 6 else ''}: {v / 1e6:.1f} ms" for k, v in sorted(conc_hist.items())))
    for n, v in sorted(thin_by.items(), key=lambda kv: -kv[1])[:18]:
        print(f"  thin {v / 1e6:7.2f} ms  {n.replace('(anonymous namespace)::', '').replace('void ', '')[:100]}")


if __name__ == "__main__":
    main()
