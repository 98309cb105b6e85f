#!/bin/bash
# same-box A/B of two experiment libraries on the 1x1 layers (ab/libfovealseg_exp_prev.so: dword-store epilogue of round 4;
# ab/libfovealseg_experiments.so: 16-byte row stores through LDS), then configs[3] and the headline step
R=${GRAFT_REPO_ROOT:-$(pwd)}
export FS_CONV_PRECISION=bf16x3
for rep in 1 2; do
  for lib in libfovealseg_exp_prev.so libfovealseg_experiments.so; do
    echo "== $lib"
    for s in 5 6; do
      FS_HIP_LIB=$R/ab/$lib python3 tools/conv_microbench.py fwd 30 $s 2>/dev/null
      FS_HIP_LIB=$R/ab/$lib python3 tools/conv_microbench.py bwd_data 30 $s 2>/dev/null
    done
    MB_SET=segformer FS_HIP_LIB=$R/ab/$lib python3 tools/conv_microbench.py fwd 20 2>/dev/null | head -13
  done
done
unset FS_CONV_PRECISION
B="bench.py --conv-precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d"
for rep in 1 2; do
  for lib in libfovealseg_exp_prev.so libfovealseg_experiments.so; do
    echo "step $lib: $(FS_HIP_LIB=$R/ab/$lib python3 $B 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    echo "config3 $lib: $(FS_HIP_LIB=$R/ab/$lib python3 tools/config_bench.py config3 16 4 bf16x3 2>/dev/null | tail -1 | cut -c1-140)"
  done
done
