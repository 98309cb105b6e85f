#!/bin/bash
# phase timeline of the 3x3 bwd-weight class kernel (A/B build with FS_BUILD_DEFINES=-DFS_WGRAD_TRACE, kept as ab/wgrad_trace.so): cycles
# per patch round and phase, with the second workgroup of a CU started 0 / 3 / 5 x 2048 cycles late; then timings without the trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for st in 0 3 5; do
  echo "== trace, FS_WGRAD_STAGGER=$st"
  FS_WGRAD_STAGGER=$st FS_HIP_LIB=$R/ab/wgrad_trace.so FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py wgrad 2 2>&1 | grep -a "wgrad trace" | awk '{k=$3" "$4" "$5} !seen[k]++' | head -4
done
for st in 0 2 3 4 6; do
  echo "== timing, FS_WGRAD_STAGGER=$st"
  FS_WGRAD_STAGGER=$st FS_HIP_LIB=$R/ab/libfovealseg_experiments.so FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py wgrad 30 2>/dev/null | head -5
done
