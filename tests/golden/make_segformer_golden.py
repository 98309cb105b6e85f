"""Golden for the SegFormer encoder from transformers 5.15.0 `SegformerModel` (the installed version):
run once in the build container:  python tests/golden/make_segformer_golden.py
Weights are name-keyed (fovealseg.weights) under the 4.46.2 key names and mapped onto the HF module."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from transformers import SegformerConfig, SegformerModel  # noqa: E402
import transformers  # noqa: E402
import fovealseg  # noqa: E402,F401
from fovealseg.weights import name_keyed_tensor  # noqa: E402
import segformer_oracle as SO  # noqa: E402

c = SegformerConfig()
c.depths, c.hidden_sizes, c.strides = [3, 6, 40, 3], [64, 128, 320, 512], [1, 2, 2, 2]
c.hidden_dropout_prob, c.attention_probs_dropout_prob = 0.3, 0.2
hf = SegformerModel(c).eval()
o = SO.OracleSegformer()
sd = {}
for k, t in o.state_dict().items():
    if k.startswith("segformer.encoder."):
        sd[SO.hf_key(k)] = name_keyed_tensor(k, t.shape)
missing, unexpected = hf.load_state_dict(sd, strict=True), None
x = torch.rand(1, 3, 80, 80, generator=torch.Generator().manual_seed(13))
with torch.no_grad():
    hs = hf(x, output_hidden_states=True).hidden_states
    size = hs[0].shape[-2:]
    cat = torch.cat([hs[0]] + [F.interpolate(h, size=size, mode="bilinear", align_corners=False) for h in hs[1:]], 1)
np.savez_compressed(os.path.join(HERE, "g13_segformer.npz"), x=x.numpy(), crop=cat[0, :, 32:48, 32:48].numpy(),
                    chan_mean=cat.mean(dim=(0, 2, 3)).numpy(), stage3=hs[3][0].numpy(), version=np.array(transformers.__version__))
print("wrote g13_segformer", cat.shape, float(cat.abs().max()))
