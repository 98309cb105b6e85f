R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05quick
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --conv-precision bf16x3 --steps 3 --warmup 1 --serial-streams --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
cp $(ls $O/stats/*/*kernel_stats.csv | tail -1) $O/bench_serial_kernel_stats_bf16x3.csv
rm -rf $O/stats
cd $R
{ python3 tools/config_bench.py config3 16 4 bf16x3 2>/dev/null | tail -1; python3 tools/config_bench.py config4 16 6 bf16x3 2>/dev/null | tail -1; } > $O/config_bench.txt
cat $O/config_bench.txt
