#!/usr/bin/env python3
"""Error of the conv kernels against an fp64 reference (max and rms, relative to the output rms), for every
precision mode, next to torch's fp32 CPU conv.  DIST=relu|heavy|grad selects the input distribution."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import fovealseg
from fovealseg import ops

g = torch.Generator().manual_seed(0)
dist = os.environ.get("DIST", "normal")


def err(y, ref):
    rms = ref.pow(2).mean().sqrt()
    return f"max {float((y - ref).abs().max() / rms):.2e} rms {float((y - ref).pow(2).mean().sqrt() / rms):.2e}"


for (B, H, C, k) in ((2, 40, 64, 3), (2, 20, 256, 3), (1, 10, 512, 3), (2, 80, 960, 3), (2, 40, 256, 1)):
    Co = 240 if C == 960 else C
    x = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    if dist == "relu":
        x = x.clamp_min(0)
    elif dist == "heavy":          # per-element log-normal magnitudes over ~6 decades
        x = x * torch.exp(3 * torch.randn(B, C, H, H, generator=g))
    elif dist == "grad":           # per-pixel magnitudes over ~8 decades, tiny absolute scale
        x = x * torch.exp(4 * torch.randn(B, 1, H, H, generator=g)) * 1e-6
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2)
    dyt = torch.randn(B, Co, H, H, generator=g)
    refdx = torch.nn.grad.conv2d_input(x.shape, w.double(), dyt.double(), 1, k // 2)
    refdw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dyt.double(), 1, k // 2)
    yc = F.conv2d(x, w, None, 1, k // 2).double()
    line = [f"C={C}->{Co} k={k} H={H} [{dist}]  torch-cpu-f32 fwd {err(yc, ref)}"]
    wd = ops.new_rsck_weight(Co, C, k, k, device="cuda")
    wd.copy_(w)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dyt.permute(0, 2, 3, 1).contiguous().cuda()
    for mode in ("f32", "bf16x3", "f16x2"):
        fovealseg.hip.set_conv_precision(mode)
        y = ops.conv2d_fwd(xd, wd, None, 1, k // 2).permute(0, 3, 1, 2).cpu().double()
        dx = ops.conv2d_bwd_data(dyd, wd, xd.shape, 1, k // 2).permute(0, 3, 1, 2).cpu().double()
        dw = ops.conv2d_bwd_weight(xd, dyd, wd.shape, 1, k // 2).cpu().double()
        line.append(f"   {mode:7s} fwd {err(y, ref)} | dgrad {err(dx, refdx)} | wgrad {err(dw, refdw)}")
    print("\n".join(line), flush=True)
