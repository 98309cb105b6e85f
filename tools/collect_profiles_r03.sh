#!/bin/bash
# Profiles committed under profiles/r03 (run on the GPU box through gpurun): rocprofv3 kernel stats of the serialised bench in the
# headline mode (bf16x3), HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, no trace domains besides --kernel-trace), SQ counters
# of the dominant kernels on their most common shapes, the DVFS probe (same kernels on all-zero operands) and configs[3] kernel stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --conv-precision bf16x3 --steps 3 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/bench_serial_kernel_stats_bf16x3.csv
rm -rf $O/stats
echo "[profiles] kernel stats done"
B1="python3 $R/bench.py --conv-precision bf16x3 --steps 1 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B1 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $B1 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/traffic_summary.py $O/fetch $O/write $O/hbm_traffic_serial.json > $O/traffic_top.txt
rm -rf $O/fetch $O/write
echo "[profiles] traffic done"
cd $R
export FS_CONV_PRECISION=bf16x3
mkdir -p $O/pmc
for spec in "fwd 0 conv3x3_wino_kernel" "fwd 3 conv3x3_wino8_kernel" "wgrad 0 conv_wgrad_class_kernel"; do
  set -- $spec
  bash tools/pmc_conv.sh $1 $2 $3 > $O/pmc/sq_bf16x3_$3_$1_shape$2.txt 2>&1
done
rm -rf $R/gpurun_out/pmc_1 $R/gpurun_out/pmc_2 $R/gpurun_out/pmc_3
echo "[profiles] SQ counters done"
# DVFS probe: the same instruction streams on all-zero operands (MI355X_MICROARCH.md, give-back (1)); random operands first
{ for z in 0 1 0 1; do for i in 0 1 2 3; do echo "zeros=$z $(MB_ZEROS=$z python3 tools/conv_microbench.py all 40 $i 2>/dev/null | tail -1)"; done; done; } > $O/dvfs_probe_zeros_vs_random.txt
unset FS_CONV_PRECISION
echo "[profiles] DVFS probe done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python3 $R/tools/config_bench.py config3 16 2 bf16x3 > $O/c3.log 2>&1 || { tail -5 $O/c3.log; exit 1; }
cp $(ls $O/c3/*/*kernel_stats.csv | tail -1) $O/config3_kernel_stats_bf16x3_b16.csv
rm -rf $O/c3
cd $R
{ python3 tools/config_bench.py config3 16 4 bf16x3 2>/dev/null | tail -1; python3 tools/config_bench.py config3 64 3 bf16x3 2>/dev/null | tail -1;
  python3 tools/config_bench.py config2 32 4 bf16x3 2>/dev/null | tail -1; python3 tools/config_bench.py config4 16 4 bf16x3 2>/dev/null | tail -1; } > $O/config_bench.txt
echo "[profiles] configs done"
ls -la $O
