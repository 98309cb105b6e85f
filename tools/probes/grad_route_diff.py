#!/usr/bin/env python3
"""Per-parameter difference between the arena-direct gradient route and the autograd (AccumulateGrad) route on a small SegFormer step,
and between two identical direct runs (the noise floor).  Usage: python tools/probes/grad_route_diff.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import ops, train


def main():
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
    dev = torch.device("cuda", 0)
    module, nets = train.build_module(cfg, device=dev)
    module.train()
    opts = train.create_optimizers(nets, cfg)
    batch = train.synthetic_batch(2, 128, 128, seed=9, device=dev)
    make_feed = lambda: {"img_data": batch[0], "seg_label": batch[2], "focus_point": batch[1], "cls_label": batch[3]}      # forward replaces seg_label

    def run(direct):
        ops.DIRECT_GRAD = direct
        for o in opts:
            o.zero_grad()
        ops.DropoutState.seed, ops.DropoutState.step = 5, 0
        ops.reset_step_state()
        loss, _, _ = module(make_feed())
        loss.mean().backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in module.named_parameters() if p.grad is not None}, float(loss.detach().mean())
    a, la = run(True)
    b, lb = run(True)
    c, lc = run(False)
    print("losses", la, lb, lc)
    rows = []
    for n in a:
        sc = float(c[n].abs().max()) + 1e-30
        rows.append((float((a[n] - c[n]).abs().max()) / sc, float((a[n] - b[n]).abs().max()) / sc, n, tuple(a[n].shape)))
    rows.sort(reverse=True)
    for r in rows[:25]:
        print(f"direct-vs-auto {r[0]:9.2e}   direct-vs-direct {r[1]:9.2e}   {r[2]} {r[3]}")


if __name__ == "__main__":
    main()
