#!/bin/bash
# same-box A/B (experiments build): F(4,3) row kernel (FS_WINO4=2: every eligible layer) against the F(2,3) kernels (FS_WINO4=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
echo "== correctness vs fp64 (FS_WINO4=2)"
FS_WINO4=2 FS_CONV_PRECISION=bf16x3 python3 tools/wino4_check.py 2>&1 | tail -12 || exit 1
FS_WINO4=2 FS_WINO4_RD=6 FS_CONV_PRECISION=bf16x3 python3 tools/wino4_check.py 2>&1 | tail -2 || exit 1
for v in "0 3" "2 3" "2 6" "0 3" "2 3" "2 6"; do
  set -- $v
  echo "== FS_WINO4=$1 FS_WINO4_RD=$2"
  for s in 0 1 2 3 4 8; do
    FS_WINO4=$1 FS_WINO4_RD=$2 FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py fwd 30 $s 2>/dev/null
    FS_WINO4=$1 FS_WINO4_RD=$2 FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py bwd_data 30 $s 2>/dev/null
  done
done
