# A/B of two python trees on the same box: $1 = tree A (git worktree), current tree = B
R=${GRAFT_REPO_ROOT:-$(pwd)}
B="bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for tree in $1 .; do
    echo "step $tree: $(cd $R/$tree && python3 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
