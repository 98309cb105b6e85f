#!/bin/bash
# same-box A/B: modules of an HRNet stage chained stream by stream, one join per stage (FS_STREAM_DEPS=1) against two joins per module (0)
B="bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for v in 0 1; do
    echo "step FS_STREAM_DEPS=$v: $(FS_STREAM_DEPS=$v python3 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
