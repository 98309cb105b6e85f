set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests/test_hip_kernels.py -m gpu -q -x --durations=15 -k "attention or grid_upsample or config3 or config4 or segformer or g7 or full_depth" > gpurun_out/r02/t3.log 2>&1; echo "pytest rc=$?"; tail -45 gpurun_out/r02/t3.log
