#!/bin/bash
# same-box A/B: bwd-weight of stride >= filter layers, one launch per single-tap class (0) vs gathered-row linear kernel (1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3
for v in 0 1 0 1; do
  for i in 9 10; do
    echo "gather=$v $(FS_WGRAD_GATHER=$v python3 tools/conv_microbench.py wgrad 20 $i 2>/dev/null | tail -1)"
  done
done
