#!/bin/bash
# same-box A/B: HRNet fuse rows on the branch streams (FS_PARALLEL_FUSE=1) against all rows on the main stream (0)
B="bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for v in 0 1; do
    echo "step FS_PARALLEL_FUSE=$v: $(FS_PARALLEL_FUSE=$v python3 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
