#!/usr/bin/env python3
"""Attribute device copies / fills / torch elementwise kernels of one training step to Python call sites."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import fovealseg
from fovealseg import train as T


def main():
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    batch = T.synthetic_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1024, 1024, seed=1, device=dev)
    T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=0)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=1)
        torch.cuda.synchronize()
    agg = collections.Counter()
    for ev in prof.events():
        n = ev.name
        if n.startswith("aten::") and any(k in n for k in ("copy_", "fill_", "zero_", "add", "clone", "contiguous", "_to_copy", "mul", "cat")):
            stack = [s for s in ev.stack if "fovealseg" in s or "foveated-instance" in s or "tools/" in s][:2]
            agg[(n, tuple(stack))] += 1
    for (n, stack), c in agg.most_common(40):
        print(c, n, " <- ".join(s.split("/")[-1] for s in stack))


if __name__ == "__main__":
    main()
