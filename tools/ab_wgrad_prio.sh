#!/bin/bash
# same-box A/B of the headline step (experiments build): issue priority rising through the bwd-weight MFMA loop (FS_WGRAD_PRIO) x split-K by
# atomics on the stride-1 layers (FS_WGRAD_STORE=1) or by per-split slabs + ordered reduce (2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for cfg in "0 1" "1 1" "1 2"; do
    set -- $cfg
    echo "FS_WGRAD_PRIO=$1 FS_WGRAD_STORE=$2: $(FS_WGRAD_PRIO=$1 FS_WGRAD_STORE=$2 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
