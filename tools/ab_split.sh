#!/bin/bash
# same-box A/B: bf16 planes by v_cvt_pk_bf16_f32 (A/B build with FS_BUILD_DEFINES=-DFS_SPLIT_CVT=1|2) against the integer split of
# the shipped library.  Conv microbench per kernel family, then three alternating pairs of the headline step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
NEW=$R/ab/libfovealseg_experiments.so
for what in fwd wgrad; do
  echo "== $what, integer split"; FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py $what 30 2>/dev/null | head -8
  echo "== $what, cvt split"; FS_HIP_LIB=$NEW FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py $what 30 2>/dev/null | head -8
done
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  echo "step, integer split: $($B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  echo "step, cvt split:     $(FS_HIP_LIB=$NEW $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
