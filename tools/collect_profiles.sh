#!/bin/bash
# Profiles committed under profiles/r01 (run on the GPU box through gpurun): rocprofv3 kernel stats of the serialised
# bench, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, no trace domains besides --kernel-trace) and SQ
# counters of the dominant kernels on their most common shape.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r01
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
B1="python3 $R/bench.py --steps 1 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B1 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $B1 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/traffic_summary.py $O/fetch $O/write $O/hbm_traffic_serial.json > $O/traffic_top.txt
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/bench_serial_kernel_stats.csv
rm -rf $O/fetch $O/write $O/stats
cd $R
for spec in "fwd 0 conv3x3_halo_kernel" "fwd 2 conv3x3_halo_kernel" "fwd 4 conv3x3_halo_kernel" "wgrad 0 conv_wgrad_class_kernel" "wgrad 4 conv_wgrad_class_kernel"; do
  set -- $spec
  bash tools/pmc_conv.sh $1 $2 $3 > $O/sq_$3_$1_shape$2.txt 2>&1
done
rm -rf $R/gpurun_out/pmc_1 $R/gpurun_out/pmc_2 $R/gpurun_out/pmc_3
ls -la $O
