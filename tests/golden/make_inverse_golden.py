"""Golden for the inverse (un-foveating) warp, SURVEY.md §8(f)-3: RUN THE REFERENCE's create_grid(..., segSize, x_inv) on CPU
(/root/reference, read-only, build container only) and store its grid, grid_inv and F.grid_sample(pred, grid_inv).
    python tests/golden/make_inverse_golden.py   ->  tests/golden/g14_inverse.npz
The nearest-neighbour hole filling of the reference (fillMissingValues_tensor, interp_mode='nearest') needs cv2, which this
image lacks, and scipy's tie choice is unspecified: that step is pinned by optimality properties in the tests instead."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402

MM = rh.load_reference()
import fovealseg  # noqa: E402,F401
from fovealseg.weights import apply_name_keyed_init  # noqa: E402

torch.set_num_threads(8)


def main():
    cfg = rh.reference_cfg()
    enc = MM.ModelBuilder.build_encoder(arch="hrnetv2_nodownsp", fc_dim=960, weights="")
    dec = MM.ModelBuilder.build_decoder(arch="c1", fc_dim=960, num_class=51, weights="")
    sal = MM.ModelBuilder.build_net_saliency(cfg=cfg, weights="")
    comp = MM.ModelBuilder.build_net_compress(cfg=cfg, weights="")
    m = MM.DeformSegmentationModule(enc, dec, sal, comp, None, cfg)
    apply_name_keyed_init(m)
    hs = ws = 80
    g = torch.Generator().manual_seed(14)
    xs_rand = torch.softmax(torch.randn(1, hs * ws, generator=g) * 2.0, 1).view(1, 1, hs, ws)
    peak = torch.full((1, hs * ws), -8.0)
    peak[0, 30 * ws + 50] = 4.0
    peak[0, 31 * ws + 50] = 3.0
    xs = torch.cat((xs_rand, torch.softmax(peak, 1).view(1, 1, hs, ws)), 0)
    pad = torch.nn.ReplicationPad2d((45, 45, 45, 45))
    seg = (96, 128)
    with torch.no_grad():
        xs_hm = pad(xs)
        grid, grid_inv = m.create_grid(xs_hm, segSize=seg, x_inv=1 - xs_hm)
        pred = torch.randn(2, 5, hs, ws, generator=g)
        unfilled = torch.isnan(grid_inv[:, :, :, 0])
        gi = grid_inv.clone()
        gi[torch.isnan(gi)] = 0
        sampled = F.grid_sample(pred, gi.float())
    out = dict(xs=xs, grid=grid, grid_inv=grid_inv, pred=pred, unfilled=unfilled, sampled=sampled, seg=np.array(seg))
    np.savez_compressed(os.path.join(HERE, "g14_inverse.npz"), **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in out.items()})
    print({k: tuple(v.shape) for k, v in out.items()}, "holes", float(unfilled.float().mean()))


if __name__ == "__main__":
    main()
