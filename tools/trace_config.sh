#!/bin/bash
# kernel trace of a configuration's training steps: busy vs span, gap histogram, top kernels.  Usage: bash tools/trace_config.sh <config> <batch> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/trace_$3
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/config_bench.py $1 $2 4 bf16x3 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
T=$(ls $O/t/*/*kernel_trace.csv | tail -1)
python3 $R/tools/trace_gaps.py $T 0.5 > $O/gaps.txt
S=$(ls $O/t/*/*kernel_stats.csv | tail -1)
cp $S $O/kernel_stats.csv
rm -rf $O/t
cat $O/gaps.txt
head -25 $O/kernel_stats.csv | cut -c1-160
