#!/usr/bin/env python3
"""In-kernel clock of the 4-wave F(2,3) kernel (bf16x3, 64->64 @ 80x80, B = 64) on random and on all-zero operands after >= 2 s of
back-to-back launches each (MI355X_MICROARCH.md, DVFS give-back (6)).  Needs ab/wino_clock.so (tools/wino_clock.sh)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FS_HIP_LIB"] = os.path.join(ROOT, "ab", "wino_clock.so")
os.environ.setdefault("FS_CONV_PRECISION", "bf16x3")
sys.path.insert(0, ROOT)
import torch
import fovealseg
from fovealseg import ops

lib = fovealseg.hip.load()
lib.fs_debug_wino_clock_ghz.restype = ctypes.c_double
for shape in ((64, 80, 80, 64, 64), (64, 40, 40, 128, 128)):
    B, H, W, Ci, Co = shape
    for zeros in (False, True, False, True):
        x = torch.zeros(B, H, W, Ci, device="cuda") if zeros else torch.randn(B, H, W, Ci, device="cuda")
        w = ops.new_rsck_weight(Co, Ci, 3, 3, device="cuda")
        w.zero_() if zeros else w.normal_()
        ops.conv2d_fwd(x, w, None, 1, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.5:
            for _ in range(200):
                ops.conv2d_fwd(x, w, None, 1, 1)
            torch.cuda.synchronize()
            n += 200
        dt = (time.perf_counter() - t0) / n
        print(f"{Ci}->{Co}@{H}x{W} {'zeros ' if zeros else 'random'}: {1e6 * dt:7.1f} us per call (kernel + pack), in-kernel clock {lib.fs_debug_wino_clock_ghz():.3f} GHz", flush=True)
