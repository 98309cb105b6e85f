#!/bin/bash
# same-box A/B of the headline step: experiments build with the round-4 strided kernels off (A) / on (B), alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  echo "A (round-4 kernel switches off): $(FS_S2FWD=0 FS_S2BWD=0 FS_WGRAD_PLANES=0 FS_WGRAD_STORE=0 FS_WGRAD_GATHER=0 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  echo "B (round-4 kernels on):       $($B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
