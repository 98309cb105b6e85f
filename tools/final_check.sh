#!/bin/bash
# Round-end evidence in one gpurun call: smoke(), the r02 profile set, the driver-style bench line.
set -o pipefail
mkdir -p gpurun_out/check
python __graft_entry__.py smoke > gpurun_out/check/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/check/smoke.log
bash tools/collect_profiles.sh > gpurun_out/check/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/check/collect.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check/bench_final.log 2>&1; echo "bench rc=$?"; tail -c 200 gpurun_out/check/bench_final.log
