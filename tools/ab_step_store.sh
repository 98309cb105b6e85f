#!/bin/bash
# same-box A/B of the headline step: bwd-weight split-K by atomics (0), store + ordered reduce on strided 3x3 layers (1), on every 3x3 layer (2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2; do
  for v in 0 1 2; do
    echo "FS_WGRAD_STORE=$v: $(FS_WGRAD_STORE=$v $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
FS_CONV_PRECISION=bf16x3 FS_WGRAD_STORE=2 python3 tools/conv_microbench.py wgrad 30 2>/dev/null | head -5
FS_CONV_PRECISION=bf16x3 FS_WGRAD_STORE=0 python3 tools/conv_microbench.py wgrad 30 2>/dev/null | head -5
