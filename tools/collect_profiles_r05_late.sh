#!/bin/bash
# Late-round refresh of the committed profiles (the kernels' epilogues changed after collect_profiles_r05.sh ran): kernel stats of the
# serialised bench in bf16x3, HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), configs[3] / configs[4] kernel stats.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05late
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
for c in "config3 16 c3" "config4 16 c4"; do
  set -- $c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$3 -- python3 $R/tools/config_bench.py $1 $2 2 bf16x3 > $O/$3.log 2>&1 || { tail -5 $O/$3.log; exit 1; }
  cp $(ls $O/$3/*/*kernel_stats.csv | tail -1) $O/${1}_kernel_stats_bf16x3_b$2.csv
  rm -rf $O/$3
done
echo "[profiles] configs done"
ls -la $O
