"""Feasibility probe: capture one configs[4] training step (DeepLab, B = 16, 2048^2) in a HIP graph and replay it.
Dropout off and learning rate / Adam step baked into the captured launches: timing and capturability only.

Outcome (round 4, ROCm 7.2 / torch 2.10.0+rocm7.0, two runs, with and without the side-stream weight gradients): the eager step runs
(28.7-29.4 ms), `torch.cuda.graph(...)` around `train.train_step` ends in a HOST segmentation fault inside the capture of the backward
(no GPU fault; torch warns about AccumulateGrad streams first).  Not pursued: a graphed step also needs the two per-step scalars that are
kernel arguments today -- dropout key, learning rate -- to come from device words."""
import os, sys, time
os.environ["FS_NAN_CHECK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import ops, train as T
which = sys.argv[1] if len(sys.argv) > 1 else "config4"
cfg = fovealseg.lvis50_cfg()
B, size = 16, 2048
if which == "config4":
    cfg.MODEL.arch_encoder = "deeplab"
elif which == "headline":
    B, size = 64, 1024
dev = torch.device("cuda", 0)
module, nets = T.build_module(cfg, device=dev)
module.train()
for m in module.modules():
    if hasattr(m, "drop_p"):
        m.drop_p = 0.0
    if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
        m.p = 0.0
opts = T.create_optimizers(nets, cfg)
batch = T.synthetic_batch(B, size, size, seed=1, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=i)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5):
    out = T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=3 + i)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 5
print(f"eager {1e3 * eager:.2f} ms/step  loss {float(out[0]):.5f}", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=8)
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    g.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / 10
print(f"replay {1e3 * rep:.2f} ms/step  loss {float(out[0]):.5f}  speed-up {eager / rep:.2f}x  -> {B / rep:.1f} img/s", flush=True)
