#!/bin/bash
# same-box A/B (experiments build): forward of stride >= filter layers on the plain kernel (FS_PW_GATHER=0) vs the 1x1 GEMM kernel over gathered rows
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
for v in 0 1 0 1; do
  echo "== FS_PW_GATHER=$v"; FS_PW_GATHER=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py fwd 30 2>/dev/null | grep "s4"; FS_PW_GATHER=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py bwd_data 30 2>/dev/null | grep "s4"
done
