#!/bin/bash
# One gpurun call: the whole -m gpu suite, then the bench line as the driver runs it.  Usage: gpurun -- bash tools/gpu_check.sh
set -o pipefail
mkdir -p gpurun_out/check
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/check/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -16 gpurun_out/check/pytest.log
[ $rc -eq 0 ] || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/check/bench.log 2>&1; echo "bench rc=$?"; tail -c 300 gpurun_out/check/bench.log
