#!/bin/bash
# configs[4] kernel stats of the tree as it is (rocprofv3 --kernel-trace --stats over 2 warm-up + 2 timed steps of tools/config_bench.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c4
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -- python3 $R/tools/config_bench.py config4 16 2 bf16x3 > $O/c4.log 2>&1 || { tail -5 $O/c4.log; exit 1; }
cp $(ls $O/c4/*/*kernel_stats.csv | tail -1) $O/config4_kernel_stats_bf16x3_b16.csv
rm -rf $O/c4
tail -1 $O/c4.log
