#!/bin/bash
# Profiles committed under profiles/r05 (run on the GPU box through gpurun): rocprofv3 kernel stats of the serialised bench in the
# headline mode (bf16x3), HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, no trace domains besides --kernel-trace), SQ counters
# of the round-4 strided kernels and of the dominant kernels, configs[3] / configs[4] kernel stats and rates.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --conv-precision bf16x3 --steps 3 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/bench_serial_kernel_stats_bf16x3.csv
rm -rf $O/stats
echo "[profiles] kernel stats done"
B1="python3 $R/bench.py --conv-precision bf16x3 --steps 1 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- $B1 > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- $B1 > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 $R/tools/traffic_summary.py $O/fetch $O/write $O/hbm_traffic_serial.json > $O/traffic_top.txt
rm -rf $O/fetch $O/write
echo "[profiles] traffic done"
cd $R
export FS_CONV_PRECISION=bf16x3
mkdir -p $O/pmc
for spec in "fwd 0 conv3x3_wino4_kernel" "fwd 1 conv3x3_wino4_kernel" "fwd 2 conv3x3_wino48_kernel" "fwd 3 conv3x3_wino8_kernel" "wgrad 0 conv_wgrad_class_kernel" "fwd 7 conv_s2fwd_kernel" "bwd_data 7 conv_s2bwd_kernel" \
            "fwd 13 conv_s2fwd_kernel" "bwd_data 13 conv_s2bwd_kernel" "wgrad 12 conv_wgrad_planes_kernel" "wgrad 9 linear_wgrad_kernel"; do
  set -- $spec
  bash tools/pmc_conv.sh $1 $2 $3 > $O/pmc/sq_bf16x3_$3_$1_shape$2.txt 2>&1
done
rm -rf $R/gpurun_out/pmc_1 $R/gpurun_out/pmc_2 $R/gpurun_out/pmc_3
echo "[profiles] SQ counters done"
python3 tools/conv_microbench.py all 20 2>/dev/null > $O/microbench_final.txt
python3 tools/shape_table.py 3 2>/dev/null > $O/shape_table_final.txt
unset FS_CONV_PRECISION
cd /tmp
for c in "config3 16 c3" "config4 16 c4"; do
  set -- $c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$3 -- python3 $R/tools/config_bench.py $1 $2 2 bf16x3 > $O/$3.log 2>&1 || { tail -5 $O/$3.log; exit 1; }
  cp $(ls $O/$3/*/*kernel_stats.csv | tail -1) $O/${1}_kernel_stats_bf16x3_b$2.csv
  rm -rf $O/$3
done
cd $R
{ python3 tools/config_bench.py config3 16 4 bf16x3 2>/dev/null | tail -1; python3 tools/config_bench.py config3 64 3 bf16x3 2>/dev/null | tail -1;
  python3 tools/config_bench.py config2 32 4 bf16x3 2>/dev/null | tail -1; python3 tools/config_bench.py config4 16 6 bf16x3 2>/dev/null | tail -1; } > $O/config_bench.txt
echo "[profiles] configs done"
ls -la $O
