#!/bin/bash
# Build ab/wino_trace.so: the library with conv_wino.hip compiled under -DFS_WINO_TRACE (wave 0 of every workgroup stamps
# s_memtime at each phase boundary; the host prints the mean phase lengths after every launch).  Run here (hipcc cross-compiles),
# then on the GPU box:  FS_HIP_LIB=$PWD/ab/wino_trace.so FS_CONV_PRECISION=bf16x3 python tools/conv_microbench.py fwd 2 <shape>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/foveated-instance-segmentation_amd
python3 $P/build.py
mkdir -p $R/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DFS_WINO_TRACE -c $P/csrc/conv_wino.hip -o /tmp/conv_wino_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/wino_trace.so $(ls $P/csrc/*.o | grep -v conv_wino.o) /tmp/conv_wino_trace.o
ls -la $R/ab/wino_trace.so
