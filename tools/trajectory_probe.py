#!/usr/bin/env python3
"""Loss trajectories of N optimisation steps on one fixed batch in every conv mode, fp32-MFMA mode twice (its run-to-run spread comes
from the order of the bwd-weight float atomics and is the noise floor any comparison between modes has to be read against)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import train


def run(mode, steps, cfg):
    fovealseg.hip.set_conv_precision(mode)
    module, nets = train.build_module(cfg, device="cuda")
    module.train()
    opts = train.create_optimizers(nets, cfg)
    batch = train.synthetic_batch(4, 256, 256, seed=11, device="cuda")
    fovealseg.ops.DropoutState.seed, fovealseg.ops.DropoutState.step = 5, 0
    losses = []
    for it in range(steps):
        out = train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=it)
        losses.append(out[0].detach().reshape(-1)[0])
    return torch.stack(losses).double().cpu()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cfg = fovealseg.lvis50_cfg()
    ref = run("f32", steps, cfg)
    print("f32     ", " ".join(f"{v:.4f}" for v in ref[::4]))
    for mode in ("f32", "bf16x3", "f16x2"):
        c = run(mode, steps, cfg)
        rel = ((c - ref).abs() / ref.abs())
        print(f"{mode:8s}", " ".join(f"{v:.4f}" for v in c[::4]), " max rel dev", f"{float(rel.max()):.2e}", "at step", int(rel.argmax()),
              " first 5 steps", " ".join(f"{float(v):.1e}" for v in rel[:5]))


if __name__ == "__main__":
    main()
