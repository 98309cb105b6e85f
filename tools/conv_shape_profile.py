#!/usr/bin/env python3
"""Per-shape conv time of one serialised training step of the bench workload (HIP events per launch).
Usage: python tools/conv_shape_profile.py [bf16x3|f32] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import modules as Mods
from fovealseg import ops
from fovealseg import train as T


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    batch_size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    fovealseg.hip.set_conv_precision(prec)
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    batch = T.synthetic_batch(batch_size, 1024, 1024, seed=1, device=dev)
    Mods.PARALLEL_BRANCHES = False
    T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=0)
    ops.TIMER = ops.KernelTimer()
    torch.cuda.synchronize()
    T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=1)
    torch.cuda.synchronize()
    shapes = ops.TIMER.by_shape()
    ops.TIMER = None
    total = sum(v[1] for v in shapes.values())
    print(f"conv total {total:.1f} ms  ({prec})")
    print("entry point            B,H,W,Cin,Ho,Wo,Cout,R,S,stride,pad,dil        n   total_ms   avg_us     TF")
    for tag, (n, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
        print(f"{tag[0]:22s} {','.join(str(a) for a in tag[1:13]):42s} {n:4d} {ms:9.2f} {1e3 * ms / n:9.1f} {fl / ms / 1e9:7.1f}")


if __name__ == "__main__":
    main()
