#!/usr/bin/env python3
"""Which Python call sites issue tensor copies during one training step (aten::copy_ / clone / contiguous / _to_copy), by
innermost frame inside this repo.  Finds the source of the ~870 __amd_rocclr_copyBuffer launches per step."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import fovealseg
from fovealseg import train as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "clone", "_to_copy", "contiguous")):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(ROOT) and "copy_census" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            shape = tuple(args[0].shape) if args and torch.is_tensor(args[0]) else ()
            self.sites[(name, site, len(shape))] += 1
        return func(*args, **(kwargs or {}))


def main():
    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device="cuda")
    module.train()
    opts = T.create_optimizers(nets, cfg)
    batch = T.synthetic_batch(8, 512, 512, seed=1, device="cuda")
    T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=0)
    with Census() as c:
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=1)
    for (name, site, nd), n in c.sites.most_common(25):
        print(f"{n:6d}  {name:28s} ndim={nd}  {site}")


if __name__ == "__main__":
    main()
