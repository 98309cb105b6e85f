#!/usr/bin/env python3
"""F(4,3) row kernel (csrc/conv_wino4.hip) against fp64 on small batches: forward (bias, BatchNorm partial sums) and bwd-data, every
plan the HRNet maps produce (80 / 40 / 20 wide), ragged channel counts.  Prints max-norm errors relative to the output maximum.
Usage: python tools/wino4_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import fovealseg
from fovealseg import ops, hip

CASES = [  # B, H, W, Cin, Cout
    (2, 80, 80, 64, 64), (3, 40, 40, 128, 128), (5, 20, 20, 256, 256), (2, 80, 80, 192, 24), (2, 16, 24, 36, 100), (1, 12, 8, 32, 16),
    (2, 80, 80, 96, 240),
    (5, 20, 20, 256, 256), (2, 40, 40, 288, 200), (1, 12, 8, 320, 72), (3, 16, 24, 256, 128),      # eight-wave form: > 64 destination, >= 256 source channels
]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    worst = 0.0
    for (B, H, W, Ci, Co) in CASES:
        x = torch.randn(B, H, W, Ci, device=dev).relu_()
        w = ops.new_rsck_weight(Co, Ci, 3, 3, device=dev)
        w.normal_(std=(9 * Ci) ** -0.5)
        bias = torch.randn(Co, device=dev)
        choice = hip.load().fs_conv2d_kernel_choice(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0, hip.conv_workspace_bytes(H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0))
        y, slab, nwg = ops.conv2d_fwd_stats(x, w, bias, 1, 1)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), bias.double().cpu(), 1, 1).permute(0, 2, 3, 1)
        e_f = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        sums = slab.view(nwg, Co, 2).double().sum(0).cpu()
        e_s = float((sums[:, 0] - ref.reshape(-1, Co).sum(0)).abs().max() / ref.reshape(-1, Co).sum(0).abs().max())
        e_q = float((sums[:, 1] - (ref.reshape(-1, Co) ** 2).sum(0)).abs().max() / (ref.reshape(-1, Co) ** 2).sum(0).abs().max())
        dy = torch.randn(B, H, W, Co, device=dev)
        dx = ops.conv2d_bwd_data(dy, w, x.shape, 1, 1)
        xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
        F.conv2d(xr, w.double().cpu(), None, 1, 1).backward(dy.permute(0, 3, 1, 2).double().cpu())
        dref = xr.grad.permute(0, 2, 3, 1)
        e_b = float((dx.double().cpu() - dref).abs().max() / dref.abs().max())
        worst = max(worst, e_f, e_b)
        print(f"B{B} {H}x{W} {Ci}->{Co}: kernel choice {choice}  fwd {e_f:.2e}  sum {e_s:.2e}  sumsq {e_q:.2e}  bwd-data {e_b:.2e}", flush=True)
    print("worst", worst)
    assert worst <= 1e-5, worst


if __name__ == "__main__":
    main()
