#!/bin/bash
# same-box A/B (experiments build): 1x1 GEMM kernel, load -> barrier -> split -> barrier -> MFMAs per chunk pair (FS_PW_DB=0) vs the
# double-buffered chunk loop (1): microbench of the stride-4 and 1x1 shapes, then configs[3] and configs[4]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
for v in 0 1; do
  echo "== FS_PW_DB=$v"
  FS_PW_DB=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py fwd 30 2>/dev/null | grep "k1 \|s4"
  FS_PW_DB=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py bwd_data 30 2>/dev/null | grep "k1 \|s4"
done
for v in 0 1 0 1; do
  echo "config3 FS_PW_DB=$v: $(FS_PW_DB=$v python3 tools/config_bench.py config3 16 12 2>/dev/null | tail -1 | cut -c1-150)"
done
for v in 0 1 0 1; do
  echo "config4 FS_PW_DB=$v: $(FS_PW_DB=$v python3 tools/config_bench.py config4 16 20 2>/dev/null | tail -1 | cut -c1-150)"
done
