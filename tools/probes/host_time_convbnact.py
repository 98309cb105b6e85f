"""Host time spent inside ops.ConvBnAct.forward / backward per training step (configs[4]: host-bound), with cProfile of the backward."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import ops, train as T
cfg = fovealseg.lvis50_cfg()
cfg.MODEL.arch_encoder = "deeplab"
dev = torch.device("cuda", 0)
module, nets = T.build_module(cfg, device=dev)
module.train()
opts = T.create_optimizers(nets, cfg)
batch = T.synthetic_batch(16, 2048, 2048, seed=1, device=dev)
for i in range(3):
    T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=i)
torch.cuda.synchronize()
acc = {"fwd": [0, 0.0], "bwd": [0, 0.0]}
F0, B0 = ops.ConvBnAct.forward, ops.ConvBnAct.backward
pr = cProfile.Profile()
def fwd(ctx, *a):
    t = time.perf_counter(); r = F0(ctx, *a); acc["fwd"][0] += 1; acc["fwd"][1] += time.perf_counter() - t; return r
def bwd(ctx, *a):
    t = time.perf_counter(); pr.enable(); r = B0(ctx, *a); pr.disable(); acc["bwd"][0] += 1; acc["bwd"][1] += time.perf_counter() - t; return r
ops.ConvBnAct.forward = staticmethod(fwd); ops.ConvBnAct.backward = staticmethod(bwd)
t0 = time.perf_counter()
T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=5)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"enqueue {1e3 * (t1 - t0):.1f} ms; ConvBnAct.forward {acc['fwd'][0]} calls {1e3 * acc['fwd'][1]:.1f} ms; backward {acc['bwd'][0]} calls {1e3 * acc['bwd'][1]:.1f} ms (with cProfile overhead)")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
