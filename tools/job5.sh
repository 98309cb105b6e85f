set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "conv_fwd_bwd or odd_input or conv_bn_act" > gpurun_out/r02/t5.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r02/t5.log
[ $rc -eq 0 ] || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02/bench_full.log 2>&1; echo "bench rc=$?"; tail -c 400 gpurun_out/r02/bench_full.log
bash tools/collect_profiles.sh > gpurun_out/r02/collect.log 2>&1; echo "collect rc=$?"; tail -15 gpurun_out/r02/collect.log
