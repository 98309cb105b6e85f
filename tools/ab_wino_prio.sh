#!/bin/bash
# same-box A/B (experiments build): F(2,3) 4-wave kernel with issue priority rising through a chunk's MFMA steps (FS_WINO_PRIO=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
for v in 0 1 0 1; do
  echo "== FS_WINO_PRIO=$v"
  FS_WINO_PRIO=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py fwd 30 2>/dev/null | head -5
  FS_WINO_PRIO=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py bwd_data 30 2>/dev/null | head -2
done
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for v in 0 1; do
    echo "step FS_WINO_PRIO=$v: $(FS_WINO_PRIO=$v $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
