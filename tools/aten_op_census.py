#!/usr/bin/env python3
"""Which ATen operators (not our C-ABI kernels) run inside one training step, and how often: torch.profiler census of one
train_step after warm-up.  Usage (GPU box): python tools/aten_op_census.py [mode]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import train as T
from fovealseg import ops


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    fovealseg.hip.set_conv_precision(mode)
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    batch = T.synthetic_batch(16, 512, 512, seed=1, device=dev)
    ops.DropoutState.seed = 1
    for i in range(2):
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=3)
        torch.cuda.synchronize()
    ev = prof.key_averages(group_by_stack_n=6)
    rows = [e for e in ev if e.key.startswith("aten::") and e.count >= 5]
    rows.sort(key=lambda e: -e.count)
    for e in rows[:40]:
        stack = " <- ".join(s.split("/")[-1] for s in e.stack[:4]) if e.stack else ""
        print(f"{e.count:6d} x {e.key:28s} cuda {e.device_time_total / 1e3:8.2f} ms   {stack[:150]}")
    print("---- memory copies: the CPU op (and python frames) that issued each one")
    cop = {}
    for e in prof.events():
        if "emcpy" in e.name or "copyBuffer" in e.name:
            par, chain = e.cpu_parent, []
            while par is not None and len(chain) < 4:
                chain.append(par.name)
                par = par.cpu_parent
            key = (e.name[:40], " <- ".join(chain))
            cop[key] = cop.get(key, 0) + 1
    for k, v in sorted(cop.items(), key=lambda kv: -kv[1])[:12]:
        print(f"{v:6d} x {k[0]:40s} {k[1][:160]}")
    for e in ev:
        if "copy" in e.key.lower() and e.count >= 50:
            stack = " <- ".join(s.split("/")[-1] for s in e.stack[:5]) if e.stack else ""
            print(f"{e.count:6d} x {e.key:28s} {stack[:200]}")
    print("---- device kernels / memcpy by name")
    agg = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            d = agg.setdefault(e.name[:70], [0, 0.0])
            d[0] += 1; d[1] += e.device_time
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:15]:
        print(f"{v[0]:6d} x {k:70s} {v[1] / 1e3:8.2f} ms")


if __name__ == "__main__":
    main()
