#!/bin/bash
# same-box A/B of two experiment libraries on the 3x3 stride-1 microbench shapes: ab_libs.sh <libA> <libB> [shapes...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; shift 2
SH=${@:-0 1 8}
for rep in 1 2; do
  for lib in $A $B; do
    echo "== $lib"
    for s in $SH; do
      FS_HIP_LIB=$R/ab/$lib FS_WINO4=2 FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py fwd 30 $s 2>/dev/null
      FS_HIP_LIB=$R/ab/$lib FS_WINO4=2 FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py bwd_data 30 $s 2>/dev/null
    done
  done
done
