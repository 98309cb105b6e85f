R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05trace
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --conv-precision bf16x3 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
f=$(ls $O/t/*/*kernel_trace.csv | tail -1)
python3 $R/tools/trace_concurrency.py $f 0.5 > $O/concurrency.txt
cat $O/concurrency.txt
head -2 $f | cut -c1-600
python3 $R/tools/trace_gaps.py $f 0.5 > $O/gaps.txt; cat $O/gaps.txt; rm -rf $O/t
