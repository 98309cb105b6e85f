#!/usr/bin/env python3
"""Which copies (blit kernels / memcpy nodes) one steady-state training step of the headline configuration issues: torch.profiler's
device-side memcpy / memset records of ONE step, grouped by kind and size, with the Python-side op that issued the most common ones.
Usage: python tools/probes/memcpy_census.py [batch]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import train as T


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = fovealseg.lvis50_cfg()
    dev = torch.device("cuda", 0)
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    data = T.synthetic_batch(batch, 1024, 1024, seed=1, device=dev)
    for i in range(3):
        T.train_step(module, opts, data, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        T.train_step(module, opts, data, cfg, epoch=1, cur_iter=3)
        torch.cuda.synchronize()
    kinds = collections.Counter()
    dur = collections.Counter()
    cpu_ops = collections.Counter()
    for ev in prof.events():
        name = ev.name
        if ev.device_type == torch.autograd.DeviceType.CUDA and ("emcpy" in name or "emset" in name or "copyBuffer" in name or "fillBuffer" in name):
            kinds[name] += 1
            dur[name] += ev.device_time
        if ev.device_type == torch.autograd.DeviceType.CPU and name.startswith("aten::") and name in (
                "aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::add_", "aten::add",
                "aten::cat", "aten::_local_scalar_dense", "aten::item", "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::slice", "aten::view", "aten::as_strided"):
            cpu_ops[name] += 1
    print("device-side copy / fill records of one step:")
    for k, n in kinds.most_common():
        print(f"  {n:5d} x {k[:100]:100s} {dur[k] / 1e3:8.3f} ms")
    print("host-side ATen ops of one step:")
    for k, n in cpu_ops.most_common():
        print(f"  {n:5d} x {k}")


if __name__ == "__main__":
    main()
