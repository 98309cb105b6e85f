#!/bin/bash
# Round-end evidence in one gpurun call: the -m gpu suite, smoke(), the r02 profile set, the driver-style bench line.
set -o pipefail
mkdir -p gpurun_out/check
python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/check/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -10 gpurun_out/check/pytest.log
[ $rc -eq 0 ] || exit 1
python __graft_entry__.py smoke > gpurun_out/check/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/check/smoke.log
bash tools/collect_profiles.sh > gpurun_out/check/collect.log 2>&1; echo "collect rc=$?"; tail -2 gpurun_out/check/collect.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check/bench_final.log 2>&1; echo "bench rc=$?"; tail -c 200 gpurun_out/check/bench_final.log
