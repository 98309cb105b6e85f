#!/usr/bin/env python3
"""Is the training-mode forward of a configuration bit-reproducible?  Runs it twice from the same dropout state with a forward hook on every
sub-module and names the first module whose output differs.  Usage: python tools/probes/forward_determinism.py [segformer|hrnetv2_nodownsp|deeplab] [size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import ops, train


def main():
    enc = sys.argv[1] if len(sys.argv) > 1 else "segformer"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    cfg = fovealseg.lvis50_cfg()
    if enc == "segformer":
        cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
    elif enc == "deeplab":
        cfg.MODEL.arch_encoder = "deeplab"
    dev = torch.device("cuda", 0)
    module, nets = train.build_module(cfg, device=dev)
    module.train()
    opts = train.create_optimizers(nets, cfg)
    batch = train.synthetic_batch(2, size, size, seed=9, device=dev)
    make_feed = lambda: {"img_data": batch[0], "seg_label": batch[2], "focus_point": batch[1], "cls_label": batch[3]}      # forward replaces seg_label
    records = []

    def digest(t):
        t = t.detach().contiguous()
        return (tuple(t.shape), int(t.view(torch.int32).to(torch.int64).sum()) if t.dtype == torch.float32 else int(t.to(torch.int64).sum()))

    def hook(name):
        def f(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            records[-1].append((name, [digest(o) for o in outs if isinstance(o, torch.Tensor)]))
        return f
    for n, m in module.named_modules():
        if n:
            m.register_forward_hook(hook(n))
    losses = []
    for rep in range(3):
        records.append([])
        for o in opts:
            o.zero_grad()
        ops.DropoutState.seed, ops.DropoutState.step = 5, 0
        ops.reset_step_state()
        loss, _, _ = module(make_feed())
        torch.cuda.synchronize()
        losses.append(float(loss.detach().mean()))
    print("losses", losses)
    for rep in (1, 2):
        first = None
        for (n0, d0), (n1, d1) in zip(records[0], records[rep]):
            assert n0 == n1
            if d0 != d1:
                first = n0
                break
        print(f"run 0 vs run {rep}: first differing module output: {first}   ({len(records[0])} module outputs compared)")


if __name__ == "__main__":
    main()
