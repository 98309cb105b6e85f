#!/bin/bash
# same-box A/B of the stride-2 bwd-data route (experiments build: FS_S2BWD = 0 four tap-class launches / 1 one launch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3
for v in 0 1 0 1; do
  for i in 7 12 13 14 15 11; do
    echo "s2bwd=$v $(FS_S2BWD=$v python3 tools/conv_microbench.py bwd_data 30 $i 2>/dev/null | tail -1)"
  done
done
