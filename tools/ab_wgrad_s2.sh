#!/bin/bash
# same-box A/B of the strided-3x3 bwd-weight kernels (experiments build: FS_WGRAD_PLANES, FS_WGRAD_S2_WGS)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3
for cfg in "0 512" "2 512" "2 256" "2 1024" "0 512" "2 512"; do
  set -- $cfg
  for i in 7 12 13 15; do
    echo "planes=$1 wgs=$2 $(FS_WGRAD_PLANES=$1 FS_WGRAD_S2_WGS=$2 python3 tools/conv_microbench.py wgrad 30 $i 2>/dev/null | tail -1)"
  done
done
