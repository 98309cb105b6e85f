#!/bin/bash
# same-box A/B: weight gradients of EVERY layer on the side stream (FS_WGRAD_SIDE_GFLOP=1000) against the small layers only (4, the default)
B="bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for v in 4 1000; do
    echo "step FS_WGRAD_SIDE_GFLOP=$v: $(FS_WGRAD_SIDE_GFLOP=$v python3 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
