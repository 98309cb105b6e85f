#!/bin/bash
# same-box A/B: bwd-weight of strided 3x3 layers, split-K atomics (0) vs per-split slabs by plain stores + one ordered reduce launch (1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3
for v in 0 1 0 1; do
  for i in 7 12 13 14 15 11; do
    echo "store=$v $(FS_WGRAD_STORE=$v python3 tools/conv_microbench.py wgrad 30 $i 2>/dev/null | tail -1)"
  done
done
