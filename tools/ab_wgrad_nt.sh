#!/bin/bash
# same-box A/B: 3x3 stride-1 bwd-weight split-K by atomics (FS_WGRAD_STORE=1, the shipped route) / plain-store slabs +
# ordered reduce (FS_WGRAD_STORE=2) / the same with NON-TEMPORAL slab stores (library built with -DFS_WGRAD_NT)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in experiments exp_nt; do
  export FS_HIP_LIB=$R/ab/libfovealseg_$lib.so
  for v in 1 2; do
    echo "== lib $lib FS_WGRAD_STORE=$v"
    for s in 0 1 2; do FS_CONV_PRECISION=bf16x3 FS_WGRAD_STORE=$v python3 tools/conv_microbench.py wgrad 30 $s 2>/dev/null; done
  done
done
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for cfg in "experiments 1" "experiments 2" "exp_nt 2"; do
    set -- $cfg
    echo "step lib $1 FS_WGRAD_STORE=$2: $(FS_HIP_LIB=$R/ab/libfovealseg_$1.so FS_WGRAD_STORE=$2 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
