#!/bin/bash
# same-box A/B of the headline step: weight packs refilled on a side stream after the optimiser step (FS_PACK_PERSIST=1, the default)
# against the pack launch in front of every conv kernel (0).  Shipped library; three alternating pairs; then configs[4] and configs[3].
B="python3 bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2 3; do
  for v in 0 1; do
    echo "headline FS_PACK_PERSIST=$v: $(FS_PACK_PERSIST=$v $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
for cfgname in config4 config3; do
  for v in 0 1 0 1; do
    echo "$cfgname FS_PACK_PERSIST=$v: $(FS_PACK_PERSIST=$v python3 tools/config_bench.py $cfgname 16 20 2>/dev/null | tail -1)"
  done
done
