#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations on one MI355X (not the headline metric; bench.py keeps configs[1]):
  config3: SegFormer encoder (fc_dim 1024), task_input_size (160,160) on the (80,80) saliency grid, 1024x1024 input
  config4: DeepLab encoder, 2048x2048 input -> (80,80) warp, batch 16 (the per-GPU share of the 8-GPU configuration)
  config2: HRNetV2, 640x640 frames, batch 32 per GPU
Usage: python tools/config_bench.py <config3|config4|config2> [batch] [steps] [mode]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import ops
from fovealseg import train as T


def main():
    which = sys.argv[1]
    cfg = fovealseg.lvis50_cfg()
    size, batch = 1024, 64
    if which == "config3":
        cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
        cfg.TRAIN.task_input_size = (160, 160)
    elif which == "config4":
        cfg.MODEL.arch_encoder = "deeplab"
        size, batch = 2048, 16
    elif which == "config2":
        size, batch = 640, 32
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else batch
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    mode = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
    fovealseg.hip.set_conv_precision(mode)
    dev = torch.device("cuda", 0)
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    opts = T.create_optimizers(nets, cfg)
    data = T.synthetic_batch(batch, size, size, seed=1, device=dev)
    ops.DropoutState.seed = 3
    for i in range(2):
        out = T.train_step(module, opts, data, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = T.train_step(module, opts, data, cfg, epoch=1, cur_iter=2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": which, "mode": mode, "batch": batch, "input": size, "task_input_size": list(cfg.TRAIN.task_input_size),
                      "img_per_s": round(batch / dt, 2), "ms_per_step": round(1e3 * dt, 1), "loss": round(float(out[0].detach()), 4),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}), flush=True)


if __name__ == "__main__":
    main()
