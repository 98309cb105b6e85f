#!/bin/bash
for rep in 1 2; do
for v in 1 0; do
echo -n "GC=$v: "
FS_AB_GC=$v python - <<'PY' 2>/dev/null | tail -1
import gc, os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch, fovealseg
from fovealseg import ops, train as T
cfg = fovealseg.lvis50_cfg(); cfg.MODEL.arch_encoder = "deeplab"
fovealseg.hip.set_conv_precision("bf16x3")
dev = torch.device("cuda", 0)
module, nets = T.build_module(cfg, device=dev); module.train()
opts = T.create_optimizers(nets, cfg)
data = T.synthetic_batch(16, 2048, 2048, seed=1, device=dev)
for i in range(3): out = T.train_step(module, opts, data, cfg, epoch=1, cur_iter=i)
if os.environ["FS_AB_GC"] == "0":
    gc.collect(); gc.freeze(); gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(12): out = T.train_step(module, opts, data, cfg, epoch=1, cur_iter=3 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
print(json.dumps({"img_per_s": round(16 / dt, 1), "ms": round(1e3 * dt, 2)}))
PY
done; done
