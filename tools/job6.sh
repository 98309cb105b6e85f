set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "conv_fwd_bwd or split_precision or conv_bn_act or g7" > gpurun_out/r02/t6.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r02/t6.log
[ $rc -eq 0 ] || exit 1
for prec in bf16x3 f16x2; do for g in 1 2; do echo "== $prec FS_WGRAD_GROUPS=$g"; FS_CONV_PRECISION=$prec FS_WGRAD_GROUPS=$g python tools/conv_microbench.py wgrad 20 2>&1 | grep -v amdgpu.ids | head -5; done; done > gpurun_out/r02/wgrad_groups_ab.txt 2>&1
cat gpurun_out/r02/wgrad_groups_ab.txt
for g in 1 2 1 2; do echo "== FS_WGRAD_GROUPS=$g"; FS_WGRAD_GROUPS=$g python bench.py --conv-precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-forward-only 2>&1 | grep -o '"value": [0-9.]*, "unit": "img/s", "n_gpus"'; done > gpurun_out/r02/wgrad_groups_bench_ab.txt 2>&1
cat gpurun_out/r02/wgrad_groups_bench_ab.txt
for g in 1 2; do echo "== f16x2 FS_WGRAD_GROUPS=$g"; FS_WGRAD_GROUPS=$g python bench.py --conv-precision f16x2 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-forward-only 2>&1 | grep -o '"value": [0-9.]*, "unit": "img/s", "n_gpus"'; done >> gpurun_out/r02/wgrad_groups_bench_ab.txt 2>&1
tail -4 gpurun_out/r02/wgrad_groups_bench_ab.txt
