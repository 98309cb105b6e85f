#!/bin/bash
# configs[4] B = 16: the Bottleneck's input through one fan-out (FS_DEEPLAB_FANOUT=1, shipped) vs read twice (0); alternating runs on one box
out=${1:-gpurun_out/deeplab_fanout_ab.txt}
: > "$out"
for rep in 1 2; do
  for v in 0 1; do
    echo -n "FS_DEEPLAB_FANOUT=$v: " >> "$out"
    FS_DEEPLAB_FANOUT=$v python tools/config_bench.py config4 16 8 2>/dev/null | tail -1 >> "$out" || exit 1
  done
done
cat "$out"
