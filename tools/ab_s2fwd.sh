#!/bin/bash
# same-box A/B of the stride-2 forward route (experiments build: FS_S2FWD = 0 tap-class kernel / 1 parity planes; FS_S2FWD_NW column tiling)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3
for v in "0 1" "1 1" "1 0" "1 2" "0 1" "1 1"; do
  set -- $v
  for i in 7 12 13 14 15 11; do
    echo "s2fwd=$1 nw=$2 $(FS_S2FWD=$1 FS_S2FWD_NW=$2 python3 tools/conv_microbench.py fwd 30 $i 2>/dev/null | tail -1)"
  done
done
