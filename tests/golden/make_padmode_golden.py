"""G17: the off-default settings of the sampler that the HIP path builds, taken by RUNNING THE REFERENCE (/root/reference, read-only)
on CPU in the build container:  python tests/golden/make_padmode_golden.py  ->  tests/golden/g17_padmodes.npz

  * TRAIN.def_saliency_pad_mode = 'reflect' / 'zero' (models/models.py:822-825): create_grid forward + d/dxs on G4's saliency maps, and the
    whole module (train mode, Dropout p = 0, B = 2, 256x256 -> 80x80, G11's inputs) -- loss, accuracy, edge loss, sampled label, grid,
    gradient norms of a few parameters;
  * MODEL.uniform_sample = 'Saliency' (config/defaults.py:69, models/models.py:816-818): the whole module under replication padding.

Inputs are not stored again: the stage test reads g4_grid.npz's `xs` / `cot`, the module test g11_e2e_train_p0.npz's x / y / focus / cls.
"""
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

GN = ("localization.fov_expand_1.weight", "localization.norm2.bias", "net_compress.conv_last.weight", "encoder.conv1.weight",
      "encoder.stage4.2.branches.3.3.conv2.weight", "decoder.conv_last.weight", "decoder.cls_net.fc.weight")


def build(cfg):
    enc = MM.ModelBuilder.build_encoder(arch="hrnetv2_nodownsp", fc_dim=960, weights="")
    dec = MM.ModelBuilder.build_decoder(arch="c1", fc_dim=960, num_class=51, weights="")
    sal = MM.ModelBuilder.build_net_saliency(cfg=cfg, weights="")
    comp = MM.ModelBuilder.build_net_compress(cfg=cfg, weights="")
    m = MM.DeformSegmentationModule(enc, dec, sal, comp, None, cfg)
    apply_name_keyed_init(m)
    return m


def main():
    out = {}
    g4 = np.load(os.path.join(HERE, "g4_grid.npz"))
    g11 = np.load(os.path.join(HERE, "g11_e2e_train_p0.npz"))
    cases = (("reflect", ""), ("zero", ""), ("replication", "Saliency"))
    for pad_mode, uniform in cases:
        cfg = rh.reference_cfg()
        cfg.TRAIN.def_saliency_pad_mode = pad_mode
        cfg.MODEL.uniform_sample = uniform
        m = build(cfg)
        tag = pad_mode if uniform == "" else "uniform"
        if uniform == "":
            # stage: the line of models/models.py:823 / :825 followed by create_grid, on G4's maps with G4's cotangent
            xs = torch.from_numpy(g4["xs"]).clone().requires_grad_(True)
            xs_hm = F.pad(xs, (45, 45, 45, 45), mode="reflect" if pad_mode == "reflect" else "constant")
            grid, _ = m.create_grid(xs_hm)
            (grid * torch.from_numpy(g4["cot"])).sum().backward()
            out[f"{tag}_grid"] = grid.detach().numpy()
            out[f"{tag}_dxs"] = xs.grad.numpy()
        # module: train mode, Dropout off (its stream is not the reference's), G11's inputs
        m.train(True)
        for d in m.modules():
            if isinstance(d, torch.nn.Dropout):
                d.p = 0.0
        feed = {"img_data": torch.from_numpy(g11["x"]), "seg_label": torch.from_numpy(g11["y"]).clone(),
                "focus_point": torch.from_numpy(g11["focus"]), "cls_label": torch.from_numpy(g11["cls"])}
        captured = {}
        orig_cg = m.create_grid

        def capture_grid(*a, _o=orig_cg, **k):
            r = _o(*a, **k)
            if "segSize" not in k and "grid" not in captured:
                captured["grid"] = r[0].detach().clone()
            return r
        m.create_grid = capture_grid
        m.zero_grad()
        loss, acc, edge = m(feed, rank=1, cur_iter=0)
        loss.mean().backward()
        named = dict(m.named_parameters())
        out[f"{tag}_outs"] = torch.stack([loss.detach(), acc.detach().float(), edge.detach()]).numpy()
        out[f"{tag}_label"] = feed["seg_label"].numpy()
        out[f"{tag}_e2e_grid"] = captured["grid"].numpy()
        out[f"{tag}_gn"] = np.array([float(named[k].grad.norm()) for k in GN], dtype=np.float64)
        print(tag, out[f"{tag}_outs"], out[f"{tag}_gn"])
    out["gn_names"] = np.array(GN)
    np.savez_compressed(os.path.join(HERE, "g17_padmodes.npz"), **out)
    print("wrote g17_padmodes", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
