#!/bin/bash
# headline step: C1's classification-branch gradient added in the mask branch's bwd-data epilogue (FS_C1_STASH=1, shipped) vs by a pass of its own (0)
out=${1:-gpurun_out/c1_stash_ab.txt}
: > "$out"
for rep in 1 2; do
  for v in 0 1; do
    echo -n "FS_C1_STASH=$v: " >> "$out"
    FS_C1_STASH=$v python bench.py --conv-precision bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> "$out" || exit 1
  done
done
cat "$out"
