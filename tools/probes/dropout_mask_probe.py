import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np, torch
import fovealseg
from fovealseg import ops
import fovealseg_oracle as O
key = ops.layer_key(11, 5)
for n, p in ((32256, 0.3), (32256, 0.5), (840, 0.3), (4096, 0.3)):
    x = torch.ones(n, device="cuda")
    y = ops.Dropout.apply(x, p, key).cpu().numpy() != 0
    k = O.dropout_keep_mask_nhwc(n, key, p)
    bad = np.nonzero(y != k)[0]
    print(n, p, "mismatches", len(bad), bad[:16], "dev keep rate", y.mean(), "oracle", k.mean())
    z = ops.GeluDropout.apply(torch.full((n,), 2.0, device="cuda"), p, key).cpu().numpy() != 0
    print("   gelu-dropout vs dropout mismatches", int((z != y).sum()))
import torch.nn.functional as F
g = torch.Generator().manual_seed(41)
for C, lead in ((64, (3, 37)), (128, (3, 37)), (132, (3, 37)), (320, (3, 37)), (512, (3, 37)), (2048, (3, 37)), (64, (1, 40003)), (320, (1, 5001))):
    x = torch.randn(*lead, C, generator=g); w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    cot = torch.randn(x.shape, generator=g)
x = torch.randn(2, 9, 7, 256, generator=g) * 2
cot = torch.randn(x.shape, generator=g)
ya = ops.GeluDropout.apply(x.cuda(), 0.3, key).cpu()
keep = torch.from_numpy(O.dropout_keep_mask_nhwc(x.numel(), key, 0.3)).view(x.shape)
a = (ya != 0); b_ = keep & (F.gelu(x) != 0)
bad = (a != b_).nonzero()
print("test replica mismatches", len(bad))
for idx in bad[:10]:
    t = tuple(int(v) for v in idx)
    print(t, "x", float(x[t]), "ya", float(ya[t]), "keep", bool(keep[t]), "cpu gelu", float(F.gelu(x)[t]), "dev gelu", float(ops.Gelu.apply(x.cuda()).cpu()[t]))
