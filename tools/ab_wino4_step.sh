#!/bin/bash
# same-box A/B (experiments build): correctness of the F(4,3) kernel vs fp64, kernel times, then the training step with / without it
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_HIP_LIB=$R/ab/libfovealseg_experiments.so
echo "== correctness vs fp64 (FS_WINO4=2)"
FS_WINO4=2 FS_CONV_PRECISION=bf16x3 python3 tools/wino4_check.py 2>&1 | tail -9 || exit 1
echo "== same cases on the F(2,3) kernels (FS_WINO4=0)"
FS_WINO4=0 FS_CONV_PRECISION=bf16x3 python3 tools/wino4_check.py 2>&1 | tail -9
for v in 0 2; do
  echo "== FS_WINO4=$v"
  for s in 0 1 2 8; do
    FS_WINO4=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py fwd 30 $s 2>/dev/null
    FS_WINO4=$v FS_CONV_PRECISION=bf16x3 python3 tools/conv_microbench.py bwd_data 30 $s 2>/dev/null
  done
done
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for v in 0 2; do
    echo "step FS_WINO4=$v: $(FS_WINO4=$v $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
