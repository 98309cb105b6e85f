#!/usr/bin/env python3
"""Per-shape table of the conv C-ABI calls of the serialised headline step (HIP events per call, branch streams off):
entry point, integer arguments, calls per step, total ms per step, us per call, algorithmic TF.
Usage: python tools/shape_table.py [steps] [kinds comma list, default conv_affine,conv_wgrad]
FS_SHAPE_CONFIG=config3|config4 and FS_SHAPE_BATCH select another BASELINE configuration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import ops, modules as Mods, train as T


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    kinds = (sys.argv[2] if len(sys.argv) > 2 else "conv_affine,conv_wgrad").split(",")
    fovealseg.hip.set_conv_precision(os.environ.get("FS_CONV_PRECISION", "bf16x3"))
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    which, size, nb = os.environ.get("FS_SHAPE_CONFIG", "headline"), 1024, int(os.environ.get("FS_SHAPE_BATCH", "64"))
    if which == "config3":
        cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
        cfg.TRAIN.task_input_size = (160, 160)
    elif which == "config4":
        cfg.MODEL.arch_encoder, size = "deeplab", 2048
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    for o in opts:
        o.flat.refresh_amax()
    batch = T.synthetic_batch(nb, size, size, seed=1, device=dev)
    Mods.PARALLEL_BRANCHES = False
    for i in range(2):
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.TIMER = timer
    for i in range(steps):
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=2 + i)
    torch.cuda.synchronize()
    ops.TIMER = None
    rows = {}
    for kind in kinds:
        for (s, e, f), tag in zip(timer.records.get(kind, []), timer.tags.get(kind, [])):
            d = rows.setdefault((kind,) + tuple(tag), [0, 0.0, 0.0])
            d[0] += 1; d[1] += s.elapsed_time(e); d[2] += f
    tot = 0.0
    print("kind entry (B,H,W,Cin,Ho,Wo,Cout,R,S,stride,pad,dil,..) | calls/step | ms/step | us/call | TF")
    for tag, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        tot += ms / steps
        print(f"{tag[0]:11s} {tag[1]:26s} {str(tag[2:14]):62s} {n // steps:4d} {ms / steps:8.3f} {1e3 * ms / n:8.1f} {fl / ms / 1e9 if ms else 0:7.1f}", flush=True)
    print(f"total {tot:.2f} ms/step over kinds {kinds}")


if __name__ == "__main__":
    main()
