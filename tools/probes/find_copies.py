"""Which host call sites issue the device-to-device copies (`__amd_rocclr_copyBuffer`) of one train step?

The serialised kernel statistics of round 4 show ~870 of them per step (3.7 us each); this probe runs one step under the torch
profiler with Python stacks and prints every operator whose device activity is a memcpy / memset, grouped by call site.

    python tools/probes/find_copies.py [--batch 16] [--config headline|config4]
"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--config", default="headline")
    args = ap.parse_args()
    import fovealseg
    from fovealseg import train as T, ops
    fovealseg.hip.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = fovealseg.lvis50_cfg()
    if args.config == "config4":
        cfg.MODEL.arch_encoder = "deeplab"
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    batch = T.synthetic_batch(args.batch, args.size, args.size, seed=1, device=dev)
    ops.DropoutState.seed = 1234
    for i in range(2):
        T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=2)
        torch.cuda.synchronize()
    evs = prof.events()
    dev_names = collections.Counter()
    sites = collections.Counter()
    for e in evs:
        kids = [k for k in getattr(e, "kernels", [])]
        hit = [k for k in kids if "emcpy" in k.name or "emset" in k.name or "copyBuffer" in k.name or "fillBuffer" in k.name]
        if not hit:
            continue
        for k in hit:
            dev_names[k.name] += 1
        stack = [s for s in (e.stack or []) if "fovealseg" in s or "foveated" in s or "bench" in s or "tools/" in s][:3]
        sites[(e.name, tuple(stack))] += len(hit)
    print("device-side names:", dev_names.most_common(10))
    for (name, stack), n in sites.most_common(25):
        print(f"{n:6d}  {name}")
        for s in stack:
            print("          ", s)
    # operators without a Python frame of ours (autograd engine threads): count them by name too
    byname = collections.Counter()
    for e in evs:
        if e.name.startswith("Memcpy") or e.name.startswith("Memset"):
            byname[e.name] += 1
    print("runtime activity records:", byname.most_common(10))


if __name__ == "__main__":
    main()
