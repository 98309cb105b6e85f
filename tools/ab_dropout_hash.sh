#!/bin/bash
# same-box A/B: forward 3x3 kernels with the dropout hash in the epilogue, previous build (one hash per element) vs this one (one per pair)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_CONV_PRECISION=bf16x3
for rep in 1 2; do
  for lib in $R/ab/prev.so $R/foveated-instance-segmentation_amd/csrc/libfovealseg_hip.so; do
    for i in 0 1 2 3; do
      echo "$(basename $lib) drop=0.3 $(FS_HIP_LIB=$lib MB_DROP=0.3 python3 tools/conv_microbench.py fwd 40 $i 2>/dev/null | tail -1)"
    done
    echo "$(basename $lib) drop=0   $(FS_HIP_LIB=$lib python3 tools/conv_microbench.py fwd 40 0 2>/dev/null | tail -1)"
  done
done
