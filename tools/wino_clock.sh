#!/bin/bash
# Build ab/wino_clock.so: the library with conv_wino.hip compiled under -DFS_WINO_CLOCK (the 4-wave F(2,3) kernel stamps s_memtime and
# s_memrealtime at its start and end into a buffer of its own; fs_debug_wino_clock_ghz() returns the median ratio = the in-kernel clock,
# MI355X_MICROARCH.md DVFS give-back (6)).  On the GPU box:  python tools/wino_clock.py
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/foveated-instance-segmentation_amd
python3 $P/build.py
mkdir -p $R/ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DFS_WINO_CLOCK -c $P/csrc/conv_wino.hip -o /tmp/conv_wino_clock.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/wino_clock.so $(ls $P/csrc/*.o | grep -v conv_wino.o) /tmp/conv_wino_clock.o
ls -la $R/ab/wino_clock.so
