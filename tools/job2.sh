set -o pipefail
mkdir -p gpurun_out/r02
python -m pytest tests/test_hip_kernels.py tests/test_ddp_gloo.py -m gpu -q -s --durations=25 -k "g9 or g11 or train_step_on_gpu or g7" > gpurun_out/r02/t2.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r02/t2.log
for prec in f16x2 bf16x3; do
  for lib in default ab/rowperm_identity.so; do
    if [ "$lib" = default ]; then unset FS_HIP_LIB; else export FS_HIP_LIB=$PWD/$lib; fi
    echo "== $prec $lib"; FS_CONV_PRECISION=$prec python tools/conv_microbench.py fwd 20 2>&1 | grep -v amdgpu.ids | head -5
  done
done > gpurun_out/r02/lds_rowperm_timing.txt 2>&1
cat gpurun_out/r02/lds_rowperm_timing.txt
unset FS_HIP_LIB
bash tools/pmc_conv.sh fwd 0 conv3x3_halo_kernel > gpurun_out/r02/pmc_halo_default.txt 2>&1
export FS_HIP_LIB=$PWD/ab/rowperm_identity.so
bash tools/pmc_conv.sh fwd 0 conv3x3_halo_kernel > gpurun_out/r02/pmc_halo_identity.txt 2>&1
grep -o "SQ_LDS_BANK_CONFLICT': [0-9.]*\|SQ_LDS_IDX_ACTIVE': [0-9.]*\|_dur': [0-9.]*" gpurun_out/r02/pmc_halo_default.txt gpurun_out/r02/pmc_halo_identity.txt
