#!/usr/bin/env python3
"""Which producers hand BatchNorm-backward a gradient WITHOUT the column sums (the layers that still run fs_bn_bwd_partial): counts by the
autograd function that returned the gradient tensor, with the layer's (M, C).  One training step of the headline configuration.
Usage: python tools/bn_partial_census.py [batch]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import hip as H, ops, train as T


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    cfg = fovealseg.lvis50_cfg()
    dev = torch.device("cuda", 0)
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    data = T.synthetic_batch(batch, 1024, 1024, seed=1, device=dev)
    T.train_step(module, opts, data, cfg, epoch=1, cur_iter=0)
    # tag every gradient tensor with the Function that returned it
    for name in dir(ops):
        cls = getattr(ops, name)
        if isinstance(cls, type) and issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and "backward" in cls.__dict__:
            orig = cls.__dict__["backward"].__func__ if isinstance(cls.__dict__["backward"], staticmethod) else cls.backward

            def wrapped(ctx, *g, _orig=orig, _n=name):
                out = _orig(ctx, *g)
                meta = getattr(ctx, "meta", None)
                for t in (out if isinstance(out, tuple) else (out,)):
                    if isinstance(t, torch.Tensor) and not hasattr(t, "_fs_src"):
                        try:
                            t._fs_src = _n
                            if isinstance(meta, dict) and _n == "ConvBnAct":
                                wsh = tuple(ctx.saved_tensors[1].shape)
                                t._fs_src_meta = "w%s stride %d fan %s" % (wsh, meta.get("stride", 0), ctx.fan is not None)
                        except Exception:
                            pass
                return out
            cls.backward = staticmethod(wrapped)
    seen = collections.Counter()
    bytes_ = collections.Counter()
    real = H.call
    cur = {}
    layers = []

    def spy(name, *args):
        if name == "fs_bn_bwd_partial":
            ints = [a for a in args if type(a) is int and a < (1 << 40)]      # ..., M, C, act, slab pointer
            small = [a for a in ints if a < (1 << 31)]
            seen[cur.get("src", "?")] += 1
            bytes_[cur.get("src", "?")] += small[-3] * small[-2] * 4 if len(small) >= 3 else 0
            layers.append((cur.get("src", "?"), small[-3] if len(small) >= 3 else 0, small[-2] if len(small) >= 3 else 0, cur.get("consumer", "?")))
        return real(name, *args)

    H.call = spy
    ops.hip.call = spy
    cba = ops.ConvBnAct.backward

    def cba_wrapped(ctx, dz, *rest):
        cur["src"] = getattr(dz, "_fs_src", "autograd-sum/unknown")
        cur["consumer"] = getattr(dz, "_fs_src_meta", "?")
        return cba(ctx, dz, *rest)
    ops.ConvBnAct.backward = staticmethod(cba_wrapped)
    T.train_step(module, opts, data, cfg, epoch=1, cur_iter=1)
    torch.cuda.synchronize()
    for k, n in seen.most_common():
        print(f"{n:4d} layers  {bytes_[k] / 2 ** 20:9.1f} MB of dz  produced by {k}")
    for src, M, C, cons in layers:
        print(f"    {src:16s} M = {M:8d} C = {C:4d}  {M * C * 4 / 2 ** 20:7.1f} MB   producer conv: {cons}")


if __name__ == "__main__":
    main()
