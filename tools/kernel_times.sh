#!/bin/bash
# Kernel-only durations of one microbench configuration: rocprofv3 --kernel-trace --stats over tools/conv_microbench.py.
# Usage (on the GPU box): [FS_* env] bash tools/kernel_times.sh <fwd|bwd_data|wgrad> <shape index> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt_$3
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$3 --output-format csv -- python3 $R/tools/conv_microbench.py $1 10 $2 > $R/gpurun_out/kt_$3.log 2>&1 || { tail -5 $R/gpurun_out/kt_$3.log; exit 1; }
python3 - "$R/gpurun_out/kt_$3" "$3" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "conv" in n or "pack" in n or "wino" in n:
            print(sys.argv[2], n.replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], round(float(r["AverageNs"]) / 1000, 1), "us")
PY
