#!/usr/bin/env python3
"""Where the ATen element-wise launches of one configs[3] training step come from: torch.profiler events of ONE step, the element-wise ATen
ops grouped by name and input shape, with the autograd node (if any) they ran under.
Usage: python tools/probes/aten_add_census.py [config3|config4|headline] [batch]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import train as T


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "config3"
    cfg = fovealseg.lvis50_cfg()
    size, batch = 1024, 16
    if which == "config3":
        cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
        cfg.TRAIN.task_input_size = (160, 160)
    elif which == "config4":
        cfg.MODEL.arch_encoder = "deeplab"
        size = 2048
    else:
        batch = 64
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else batch
    dev = torch.device("cuda", 0)
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    data = T.synthetic_batch(batch, size, size, seed=1, device=dev)
    for i in range(2):
        T.train_step(module, opts, data, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        T.train_step(module, opts, data, cfg, epoch=1, cur_iter=2)
        torch.cuda.synchronize()
    evs = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU], key=lambda e: e.time_range.start)
    nodes = [e for e in evs if e.name.startswith("autograd::engine::evaluate_function") or e.name.endswith("Backward") or "Backward" in e.name]
    watch = ("aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::copy_", "aten::fill_", "aten::zero_", "aten::sum", "aten::div", "aten::div_",
             "aten::cat", "aten::clone", "aten::contiguous", "aten::where", "aten::sub", "aten::neg")
    groups = collections.Counter()
    for e in evs:
        if e.name not in watch:
            continue
        owner = "(forward / host code)"
        for n in nodes:
            if n.time_range.start <= e.time_range.start and e.time_range.end <= n.time_range.end and n.name.startswith("autograd::engine::evaluate_function"):
                owner = n.name.replace("autograd::engine::evaluate_function: ", "")
                break
        shapes = str([tuple(s) for s in (e.input_shapes or []) if s])[:80]
        groups[(e.name, owner, shapes)] += 1
    for (name, owner, shapes), n in groups.most_common(40):
        print(f"{n:5d} x {name:16s} under {owner:40s} {shapes}")


if __name__ == "__main__":
    main()
