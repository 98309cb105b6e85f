#!/bin/bash
# same-box A/B of the current tree against the round-4 final tree (git worktree ab/base_r4 at b1d3172 with its own library):
# configs[3] / configs[4] rates and the headline step
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for tree in ab/base_r4 .; do
    for c in "config3 16 4" "config4 16 8"; do
      echo "$tree $c: $(cd $R/$tree && python3 tools/config_bench.py $c bf16x3 2>/dev/null | tail -1 | cut -c1-160)"
    done
  done
done
B="bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2; do
  for tree in ab/base_r4 .; do
    echo "step $tree: $(cd $R/$tree && python3 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
