set -o pipefail
mkdir -p gpurun_out/r02
for rev in 0 1; do echo "== FS_BN_REVERSE=$rev"; FS_BN_REVERSE=$rev python tools/bn_microbench.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r02/bn_reverse_ab.txt 2>&1
cat gpurun_out/r02/bn_reverse_ab.txt
for rev in 0 1 0 1; do echo "== FS_BN_REVERSE=$rev"; FS_BN_REVERSE=$rev python bench.py --conv-precision bf16x3 --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-forward-only 2>&1 | grep -o '"value": [0-9.]*, "unit": "img/s", "n_gpus"'; done > gpurun_out/r02/bn_reverse_bench_ab.txt 2>&1
cat gpurun_out/r02/bn_reverse_bench_ab.txt
python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r02/t4.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/r02/t4.log
