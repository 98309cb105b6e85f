#!/usr/bin/env python3
"""Error of the conv kernels against an fp64 reference (max and rms, relative to the output rms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import fovealseg
from fovealseg import ops
g = torch.Generator().manual_seed(0)
for (B, H, C, k) in ((2, 40, 64, 3), (2, 20, 256, 3), (1, 10, 512, 3), (2, 40, 256, 1)):
    x = torch.randn(B, C, H, H, generator=g); w = torch.randn(C, C, k, k, generator=g) / (C * k * k) ** 0.5
    x = x.abs() if os.environ.get("POS") else x          # post-ReLU-like (non-negative) activations
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2)
    wd = ops.new_rsck_weight(C, C, k, k, device="cuda"); wd.copy_(w)
    y = ops.conv2d_fwd(x.permute(0, 2, 3, 1).contiguous().cuda(), wd, None, 1, k // 2).permute(0, 3, 1, 2).cpu().double()
    yc = F.conv2d(x, w, None, 1, k // 2).double()
    rms = ref.pow(2).mean().sqrt()
    print(f"C={C} k={k} H={H}: hip max {float((y-ref).abs().max()/rms):.3e} rms {float((y-ref).pow(2).mean().sqrt()/rms):.3e} mean {float((y-ref).mean()/rms):+.3e} | torch-cpu max {float((yc-ref).abs().max()/rms):.3e} rms {float((yc-ref).pow(2).mean().sqrt()/rms):.3e}")
