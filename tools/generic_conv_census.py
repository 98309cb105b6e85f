#!/usr/bin/env python3
"""Which convolution launches of one training step fall to the generic (unaligned-channel, fp32 MFMA) kernels: shapes and counts.
Usage: python tools/generic_conv_census.py <config3|config4|config2|headline> [batch]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import hip as H, ops, train as T


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
    elif which == "config2":
        size, batch = 640, 32
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else batch
    dev = torch.device("cuda", 0)
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    data = T.synthetic_batch(batch, size, size, seed=1, device=dev)
    T.train_step(module, opts, data, cfg, epoch=1, cur_iter=0)
    seen = collections.Counter()
    real = H.call

    def spy(name, *args):
        if name.startswith("fs_conv2d"):
            ints = tuple(a for a in args if type(a) is int and 0 <= a < (1 << 20))
            # (B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ...)
            if len(ints) >= 9 and (ints[3] % 4 or ints[6] % 4):
                seen[(name,) + ints[:12]] += 1
        return real(name, *args)

    H.call = spy
    ops.hip.call = spy
    T.train_step(module, opts, data, cfg, epoch=1, cur_iter=1)
    torch.cuda.synchronize()
    for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(n, k)


if __name__ == "__main__":
    main()
