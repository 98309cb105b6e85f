#!/bin/bash
# Profiles committed under profiles/r02 (run on the GPU box through gpurun): rocprofv3 kernel stats of the serialised bench in the
# headline mode (bf16x3) and in f16x2, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, no trace domains besides
# --kernel-trace) and SQ counters of the dominant kernels on their most common shape.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for MODE in bf16x3 f16x2; do
  BENCH="python3 $R/bench.py --conv-precision $MODE --steps 3 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$MODE -- $BENCH > $O/stats_$MODE.log 2>&1 || { tail -5 $O/stats_$MODE.log; exit 1; }
  cp $(ls $O/stats_$MODE/*/*kernel_stats.csv | tail -1) $O/bench_serial_kernel_stats_$MODE.csv
  rm -rf $O/stats_$MODE
done
B1="python3 $R/bench.py --conv-precision bf16x3 --steps 1 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B1 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $B1 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/traffic_summary.py $O/fetch $O/write $O/hbm_traffic_serial.json > $O/traffic_top.txt
rm -rf $O/fetch $O/write
cd $R
export FS_CONV_PRECISION=bf16x3
for spec in "fwd 0 conv3x3_wino_kernel" "fwd 3 conv3x3_wino_kernel" "wgrad 0 conv_wgrad_class_kernel" "fwd 5 conv1x1_gemm_kernel"; do
  set -- $spec
  bash tools/pmc_conv.sh $1 $2 $3 > $O/sq_bf16x3_$3_$1_shape$2.txt 2>&1
done
FS_WINOGRAD=0 bash tools/pmc_conv.sh fwd 0 conv3x3_halo_kernel > $O/sq_bf16x3_conv3x3_halo_kernel_fwd_shape0.txt 2>&1
export FS_CONV_PRECISION=f16x2
bash tools/pmc_conv.sh fwd 0 conv3x3_halo_kernel > $O/sq_f16x2_conv3x3_halo_kernel_fwd_shape0.txt 2>&1
bash tools/pmc_conv.sh fwd 1 conv3x3_wino_kernel > $O/sq_f16x2_conv3x3_wino_kernel_fwd_shape1.txt 2>&1
unset FS_CONV_PRECISION
# kernel-only durations of the 3x3 stride-1 forward on the four dominant shapes, direct (halo) form against the F(2,3) row kernel
for m in bf16x3 f16x2; do for w in 0 1; do for s in 0 1 2 3; do
  FS_CONV_PRECISION=$m FS_WINOGRAD=$w bash tools/kernel_times.sh fwd $s ${m}_winograd${w}_shape$s
done; done; done > $O/winograd_kernel_times.txt 2>&1
rm -rf $R/gpurun_out/kt_*
# per-layer error against fp64 of the three modes, with the F(2,3) row kernel (default) and with the direct form only
{ echo "# python tools/conv_accuracy.py  (FS_WINOGRAD=1, the default: 3x3 stride-1 layers on conv3x3_wino_kernel)"; python3 tools/conv_accuracy.py 2>&1 | grep -v amdgpu.ids;
  echo "# FS_WINOGRAD=0 python tools/conv_accuracy.py  (direct form: conv3x3_halo_kernel)"; FS_WINOGRAD=0 python3 tools/conv_accuracy.py 2>&1 | grep -v amdgpu.ids; } > $O/conv_accuracy_winograd.txt
rm -rf $R/gpurun_out/pmc_1 $R/gpurun_out/pmc_2 $R/gpurun_out/pmc_3
ls -la $O
