#!/bin/bash
# same-box A/B of the tap-class kernel's column tiling (experiments build: FS_TAPSET_NW = 0 / 1 / 2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3
for nw in 0 1 2 0 1; do
  for i in 7 13 14 15 11; do
    echo "nw=$nw $(FS_TAPSET_NW=$nw python3 tools/conv_microbench.py all 30 $i 2>/dev/null | tail -1)"
  done
done
