#!/bin/bash
# same-box A/B (experiments build): eight-wave F(4,3) kernel (FS_WINO48=1: > 64 destination and >= 256 source channels; 2: every layer
# above 64 destination channels) against the four-wave F(4,3) kernel (FS_WINO48=0) and the F(2,3) kernels (FS_WINO4=0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
export FS_CONV_PRECISION=bf16x3
echo "== correctness vs fp64 (FS_WINO4=2 FS_WINO48=1)"
FS_WINO4=2 FS_WINO48=1 python3 tools/wino4_check.py 2>&1 | grep -v amdgpu | tail -13 || exit 1
echo "== correctness vs fp64 (FS_WINO4=2 FS_WINO48=2)"
FS_WINO4=2 FS_WINO48=2 python3 tools/wino4_check.py 2>&1 | tail -1 || exit 1
for rep in 1 2; do
for v in "0 0" "2 0" "2 1" "2 2"; do
  set -- $v
  echo "== FS_WINO4=$1 FS_WINO48=$2"
  for s in 1 2 4; do
    FS_WINO4=$1 FS_WINO48=$2 python3 tools/conv_microbench.py fwd 30 $s 2>/dev/null
    FS_WINO4=$1 FS_WINO48=$2 python3 tools/conv_microbench.py bwd_data 30 $s 2>/dev/null
  done
done
done
