#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the dominant HRNet shapes (B=64). Prints TFLOP/s per shape.
Usage: python tools/conv_microbench.py [fwd|bwd_data|wgrad|all] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import ops

SHAPES = [  # B, H, W, Cin, Cout, k, stride
    (64, 80, 80, 64, 64, 3, 1),
    (64, 40, 40, 128, 128, 3, 1),
    (64, 20, 20, 256, 256, 3, 1),
    (64, 10, 10, 512, 512, 3, 1),
    (64, 80, 80, 960, 240, 3, 1),
    (64, 80, 80, 64, 256, 1, 1),
    (64, 80, 80, 256, 64, 1, 1),
    (64, 80, 80, 64, 128, 3, 2),
    (64, 80, 80, 192, 192, 3, 1),
    (64, 80, 80, 960, 512, 3, 4),     # C1 classification head: stride > filter reach, every tap class is a single tap
    (64, 80, 80, 960, 512, 1, 4),
    (64, 20, 20, 512, 512, 3, 2),
    (64, 80, 80, 64, 64, 3, 2),       # [12..15] the HRNet fuse down-paths (round 4)
    (64, 40, 40, 128, 256, 3, 2),
    (64, 40, 40, 64, 256, 3, 2),
    (64, 20, 20, 256, 512, 3, 2),
]

# the linear layers of configs[3] (SegFormer-B5 at 160x160, B=16) as 1x1 convs over (1, tokens, 1, C): MB_SET=segformer
SEGFORMER = [   # strides (1,2,2,2): 160x160, 80x80, 40x40, 20x20 tokens per image
    (16, 160, 160, 64, 64, 1, 1), (16, 160, 160, 64, 256, 1, 1), (16, 160, 160, 256, 64, 1, 1),
    (16, 80, 80, 128, 128, 1, 1), (16, 80, 80, 128, 512, 1, 1), (16, 80, 80, 512, 128, 1, 1),
    (16, 40, 40, 320, 320, 1, 1), (16, 20, 20, 320, 320, 1, 1), (16, 40, 40, 320, 1280, 1, 1), (16, 40, 40, 1280, 320, 1, 1),
    (16, 20, 20, 512, 512, 1, 1), (16, 20, 20, 512, 2048, 1, 1), (16, 20, 20, 2048, 512, 1, 1),
]
if os.environ.get("MB_SET") == "segformer":
    SHAPES = SEGFORMER


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    dev = "cuda"
    for si, (B, H, W, Ci, Co, k, s) in enumerate(SHAPES):
        if only >= 0 and si != only:
            continue
        x = torch.randn(B, H, W, Ci, device=dev)
        w = ops.new_rsck_weight(Co, Ci, k, k, device=dev)
        w.normal_()
        if os.environ.get("MB_ZEROS") == "1":          # DVFS probe: same instruction stream on all-zero operands (MI355X_MICROARCH.md, give-back (1))
            x.zero_(); w.zero_()
        pad = k // 2
        y = ops.conv2d_fwd(x, w, None, s, pad)
        dy = torch.zeros_like(y) if os.environ.get("MB_ZEROS") == "1" else torch.randn_like(y)
        flops = 2.0 * y.numel() * Ci * k * k
        drop = float(os.environ.get("MB_DROP", "0"))       # forward with the dropout hash in the epilogue (BasicBlock convs: p = 0.3)
        fns = {"fwd": lambda: ops.conv2d_fwd(x, w, None, s, pad, drop, 12345 if drop > 0 else 0),
               "bwd_data": lambda: ops.conv2d_bwd_data(dy, w, x.shape, s, pad),
               "wgrad": lambda: ops.conv2d_bwd_weight(x, dy, w.shape, s, pad)}
        res = []
        for name, fn in fns.items():
            if which != "all" and which != name:
                continue
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / reps
            res.append(f"{name} {us:8.1f} us {flops / us / 1e6:6.1f} TF")
        print(f"[{si}] B{B} {H}x{W} {Ci}->{Co} k{k} s{s}: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
