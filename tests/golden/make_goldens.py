"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (/root/reference, read-only) on CPU.

Run once in the build container:  python tests/golden/make_goldens.py
The fixtures are data (inputs + the reference's outputs); weights are never stored -- they are
regenerated on both sides by fovealseg.weights.name_keyed_tensor.  Golden IDs follow SURVEY.md §8(c).
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


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def gen(seed):
    return torch.Generator().manual_seed(seed)


def synth_batch(B, H, W, seed=1):
    """SURVEY.md §8(d) synthetic inputs: uniform image, gaze in [0.1,0.9), disc mask at the gaze."""
    g = gen(seed)
    X = torch.rand(B, 3, H, W, generator=g)
    Fp = torch.rand(B, 2, generator=g) * 0.8 + 0.1
    cls = torch.randint(0, 50, (B, 1), generator=g)
    ii = torch.arange(H, dtype=torch.float32)[None, :, None]
    jj = torch.arange(W, dtype=torch.float32)[None, None, :]
    cy = (Fp[:, 0] * (H - 1))[:, None, None]
    cx = (Fp[:, 1] * (W - 1))[:, None, None]
    Y = (((ii - cy) ** 2 + (jj - cx) ** 2) <= (0.15 * H) ** 2).float().unsqueeze(1)
    return X, Fp, Y, cls


def smooth_image(B, H, W, seed):
    """Low-frequency image so end-to-end tolerances are not dominated by 0.01-px sampling jitter."""
    g = gen(seed)
    base = torch.rand(B, 3, 9, 9, generator=g)
    return F.interpolate(base, size=(H, W), mode="bicubic", align_corners=True).clamp(0, 1).contiguous()


def build_reference_module(cfg):
    enc = MM.ModelBuilder.build_encoder(arch="hrnetv2_nodownsp", fc_dim=960, weights="")
    dec = MM.ModelBuilder.build_decoder(arch="c1", fc_dim=960, num_class=51, weights="")
    sal = MM.ModelBuilder.build_net_saliency(cfg=cfg, weights="")
    comp = MM.ModelBuilder.build_net_compress(cfg=cfg, weights="")
    m = MM.DeformSegmentationModule(enc, dec, sal, comp, None, cfg)
    apply_name_keyed_init(m)
    return m


def main():
    cfg = rh.reference_cfg()
    m = build_reference_module(cfg)
    hs = ws = 80

    # ---- G1: constants ------------------------------------------------------------------
    save("g1_constants", gaussian=MM.makeGaussian(91, fwhm=45), filter=m.filter.weight[0, 0],
         p_basis=m.P_basis)

    # ---- G2: gaze map + low-res input ---------------------------------------------------
    for H in (128, 640):
        X, Fp, Y, cls = synth_batch(2, H, H, seed=10 + H)
        HS, WS = hs, ws
        max_dist = np.sqrt(HS ** 2 + WS ** 2)
        hidx = Fp[:, 0] * (HS - 1)
        widx = Fp[:, 1] * (WS - 1)
        gm = MM.gen_grid_mtx_2xHxW(HS, WS).unsqueeze(0).repeat(2, 1, 1, 1)
        dist = torch.sqrt((gm[:, 0] - hidx[:, None, None]) ** 2 + (gm[:, 1] - widx[:, None, None]) ** 2)
        focus = (dist / max_dist).unsqueeze(1) ** 2
        x_low = MM.b_imresize(X, (HS, WS), interp="bilinear")
        x_low = torch.cat((x_low, focus, focus), 1)
        save(f"g2_lowres_{H}", seed=10 + H, focus=Fp, x_low=x_low,
             x=X if H == 128 else np.zeros(0, np.float32))

    # ---- G3: saliency net + compress + softmax (eval and train) -------------------------
    x_low = torch.rand(3, 5, hs, ws, generator=gen(3))
    out = {}
    for mode in ("eval", "train"):
        m.train(mode == "train")
        with torch.no_grad():
            s = m.net_compress(m.localization(x_low))
            xs = torch.nn.Softmax(dim=1)(s.view(-1, hs * ws)).view(-1, 1, hs, ws)
        out["logit_" + mode] = s
        out["xs_" + mode] = xs
    apply_name_keyed_init(m)            # train-mode BN moved the running stats: restore
    save("g3_saliency", x_low=x_low, **out)

    # ---- G4: create_grid fwd + d/dxs -----------------------------------------------------
    g = gen(4)
    xs_rand = torch.softmax(torch.randn(2, hs * ws, generator=g) * 2.0, 1).view(2, 1, hs, ws)
    peak = torch.full((2, hs * ws), -8.0)
    peak[0, 30 * ws + 50] = 4.0
    peak[0, 31 * ws + 50] = 3.0
    peak[1, 5 * ws + 3] = 6.0
    xs_peak = torch.softmax(peak, 1).view(2, 1, hs, ws)
    xs_in = torch.cat((xs_rand, xs_peak), 0).clone().requires_grad_(True)
    pad = torch.nn.ReplicationPad2d((45, 45, 45, 45))
    grid, grid_y = m.create_grid(pad(xs_in))
    cot = torch.randn(grid.shape, generator=g)
    (grid * cot).sum().backward()
    save("g4_grid", xs=xs_in, grid=grid, grid_y=grid_y, cot=cot, dxs=xs_in.grad)

    # ---- G5: grid_sample fwd (bit-exact), label maps, bwd wrt grid -----------------------
    for (H, W) in ((128, 128), (200, 136)):
        X, Fp, Y, cls = synth_batch(2, H, W, seed=50 + H)
        gg = gen(5 + H)
        gr = (torch.rand(2, hs, ws, 2, generator=gg) * 2.2 - 1.1)          # includes out-of-range
        gr[0, 0, :8, 0] = torch.tensor([-1.0, 1.0, -1.0, 1.0, 0.0, 0.5, -0.5, 0.999])
        gr[0, 0, :8, 1] = torch.tensor([-1.0, 1.0, 1.0, -1.0, 0.0, -1.0, 1.0, -0.999])
        gr = gr.clone().requires_grad_(True)
        xs_ = F.grid_sample(X, gr)
        ys_ = F.grid_sample(Y.float(), gr).squeeze(1)
        cot = torch.randn(xs_.shape, generator=gg)
        (xs_ * cot).sum().backward()
        save(f"g5_gridsample_{H}x{W}", x=X, y=Y, grid=gr, x_sampled=xs_, y_sampled=ys_,
             label=ys_.long(), cot=cot, dgrid=gr.grad)

    # ---- G6: inverse index maps ----------------------------------------------------------
    with torch.no_grad():
        xs_hm = pad(xs_in.detach())
        for (H, W) in ((128, 128), (640, 640)):
            gfwd, ginv = m.create_grid(xs_hm, segSize=(H, W), x_inv=1 - xs_hm)
            u = (((gfwd[..., 0] + 1) / 2) * (W - 1)).int().long()
            v = (((gfwd[..., 1] + 1) / 2) * (H - 1)).int().long()
            nanmask = torch.isnan(ginv[:, :, :, 0])
            save(f"g6_inverse_{H}", grid=gfwd, u=u, v=v, nan_count=nanmask.sum(dim=(1, 2)),
                 nanmask=np.packbits(nanmask.numpy()))

    # ---- G7: HRNet building blocks fwd+bwd ----------------------------------------------
    enc = m.encoder
    blocks = {
        "basic": (enc.stage2[0].branches[0][0], "encoder.stage2.0.branches.0.0", [(2, 64, 20, 20)]),
        "bottleneck": (enc.layer1[0], "encoder.layer1.0", [(2, 64, 20, 20)]),
        "hrmodule4": (enc.stage4[0], "encoder.stage4.0", [(2, 64, 16, 16), (2, 128, 8, 8), (2, 256, 4, 4), (2, 512, 2, 2)]),
    }
    for name, (blk, prefix, shapes) in blocks.items():
        for mode in ("eval", "train_p0"):
            apply_name_keyed_init(m)
            blk.train(mode != "eval")
            for d in blk.modules():
                if isinstance(d, torch.nn.Dropout):
                    d.p = 0.0 if mode == "train_p0" else 0.3
            gg = gen(70)
            ins = [torch.randn(s, generator=gg).requires_grad_(True) for s in shapes]
            outs = blk(ins[0]) if len(ins) == 1 else blk(list(ins))
            outs = [outs] if isinstance(outs, torch.Tensor) else list(outs)
            cots = [torch.randn(o.shape, generator=gg) for o in outs]
            blk.zero_grad()
            sum((o * c).sum() for o, c in zip(outs, cots)).backward()
            arrs = {}
            for i, t in enumerate(ins):
                arrs[f"in{i}"] = t
                arrs[f"din{i}"] = t.grad
            for i, (o, c) in enumerate(zip(outs, cots)):
                arrs[f"out{i}"] = o
                arrs[f"cot{i}"] = c
            for pn, p in blk.named_parameters():
                if p.grad is not None and pn.endswith("conv1.weight"):
                    # full tensor when small, else a 16x16 (out,in) corner
                    arrs["dw:" + pn] = p.grad if p.grad.numel() <= 40000 else p.grad[:16, :16]
                if p.grad is not None and pn.endswith("bn1.weight"):
                    arrs["dgamma:" + pn] = p.grad
            save(f"g7_{name}_{mode}", prefix=np.array(prefix), **arrs)
    for d in m.modules():
        if isinstance(d, torch.nn.Dropout):
            d.p = 0.3
    apply_name_keyed_init(m)

    # ---- G8: full HRNet eval forward ----------------------------------------------------
    m.eval()
    x80 = torch.rand(1, 3, 80, 80, generator=gen(8))
    with torch.no_grad():
        feat = m.encoder(x80, return_feature_maps=True)[0]
    save("g8_hrnet_eval", x=x80, crop=feat[0, :, 32:48, 32:48], chan_mean=feat.mean(dim=(0, 2, 3)),
         chan_absmean=feat.abs().mean(dim=(0, 2, 3)), total=feat.double().sum())

    # ---- G9: C1 fwd+bwd ------------------------------------------------------------------
    for mode in ("eval", "train"):
        apply_name_keyed_init(m)
        m.decoder.train(mode == "train")
        gg = gen(9)
        f9 = (torch.randn(2, 960, 80, 80, generator=gg) * 0.5).requires_grad_(True)
        pred = m.decoder([f9])
        cot = torch.randn(pred.shape, generator=gg) * 0.01
        m.decoder.zero_grad()
        (pred * cot).sum().backward()
        save(f"g9_c1_{mode}", seed=9, pred_ch0=pred[:, :50, 0, 0], pred_last=pred[:, 50],
             dfeat_crop=f9.grad[:, ::60, 20:36, 20:36], dfeat_total=f9.grad.double().abs().sum(),
             dw_conv_last=m.decoder.conv_last.weight.grad, dfc=m.decoder.cls_net.fc.weight.grad,
             dcbr_crop=m.decoder.cbr[0].weight.grad[:8, :8])
    apply_name_keyed_init(m)

    # ---- G10: losses + accuracies --------------------------------------------------------
    gg = gen(10)
    pred = (torch.randn(3, 51, 40, 40, generator=gg) * 2).requires_grad_(True)
    gt = torch.randint(0, 51, (3, 40, 40), generator=gg)
    gt[gt < 45] = 50
    gt[1, 5:20, 5:25] = 7
    fl = MM.FocalLoss(gamma=5.0)(pred, gt)
    dl = m.crit(pred, gt)
    (fl + dl).backward()
    accs = [m.pixel_acc(pred, gt), m.fg_bin_pixel_acc(pred, gt), m.fbg_cls_pixel_acc(pred, gt),
            m.fbg_bin_pixel_acc(pred, gt)]
    xs_e = torch.softmax(torch.randn(3, 6400, generator=gg), 1).view(3, 1, 80, 80).requires_grad_(True)
    _, _, Y, _ = synth_batch(3, 256, 256, seed=11)
    t = F.interpolate(Y, size=(80, 80), mode="area")
    a = (xs_e - xs_e.min()) / (xs_e.max() - xs_e.min())
    b = (t - t.min()) / (t.max() - t.min())
    el = 0.05 * torch.nn.MSELoss()(a, b) * 100.0
    el.backward()
    save("g10_losses", pred=pred, gt=gt, focal=fl, dice=dl, dpred=pred.grad, accs=torch.stack(accs),
         xs=xs_e, y_seed=11, area=t, edge=el, dxs=xs_e.grad)

    # ---- G11: end to end, B=2, H=256 -----------------------------------------------------
    for mode in ("eval", "train_p0"):
        apply_name_keyed_init(m)
        m.train(mode != "eval")
        for d in m.modules():
            if isinstance(d, torch.nn.Dropout):
                d.p = 0.0
        X = smooth_image(2, 256, 256, seed=111)
        _, Fp, Y, cls = synth_batch(2, 256, 256, seed=112)
        feed = {"img_data": X, "seg_label": Y.clone(), "focus_point": Fp, "cls_label": cls}
        m.zero_grad()
        captured = {}
        orig_cg = m.create_grid

        def capture_grid(*a, _o=orig_cg, **k):
            out = _o(*a, **k)
            if "segSize" not in k and "grid" not in captured:
                captured["grid"] = out[0].detach().clone()
            return out
        m.create_grid = capture_grid
        if mode == "eval":
            with torch.no_grad():
                outs = m(feed, is_inference=True, rank=1, cur_iter=0)
            save("g11_e2e_eval", x=X, y=Y, focus=Fp, cls=cls, outs=torch.stack([o.float() for o in outs]),
                 label=feed["seg_label"], grid=captured["grid"])
        else:
            loss, acc, edge = m(feed, rank=1, cur_iter=0)
            loss.mean().backward()
            gn = {k: p.grad.norm() for k, p in m.named_parameters()
                  if k in ("localization.fov_expand_1.weight", "encoder.conv1.weight",
                           "decoder.conv_last.weight", "net_compress.conv_last.weight",
                           "encoder.stage4.2.branches.3.3.conv2.weight", "decoder.cls_net.fc.weight",
                           "encoder.layer1.0.conv2.weight", "encoder.stage2.0.fuse_layers.1.0.0.0.weight",
                           "encoder.stage3.2.branches.1.1.bn2.weight", "encoder.transition2.2.0.0.weight",
                           "decoder.cbr.0.weight", "decoder.cls_net.layer2.0.conv1.0.bias",
                           "localization.norm2.bias")}
            save("g11_e2e_train_p0", x=X, y=Y, focus=Fp, cls=cls, outs=torch.stack([loss, acc, edge]),
                 label=feed["seg_label"], gn_names=np.array(list(gn.keys())),
                 gn=torch.stack(list(gn.values())), grid=captured["grid"])
        del m.create_grid
    for d in m.modules():
        if isinstance(d, torch.nn.Dropout):
            d.p = 0.3

    # ---- G12: LR schedule ----------------------------------------------------------------
    import types
    tds = types.ModuleType("tds")
    src = open(os.path.join(rh.REF, "train_deform_semantic.py")).read()
    start = src.index("def adjust_learning_rate")
    end = src.index("def main(")
    exec(compile(src[start:end], "adjust_lr", "exec"), tds.__dict__)   # run the reference's own function
    params = [torch.nn.Parameter(torch.zeros(1)) for _ in range(4)]
    opts = [torch.optim.Adam([{"params": [p], "lr_mult": 0.001, "zoom": z}], lr=2e-5)
            for p, z in zip(params, (False, False, True, True))]
    table = []
    for ep in (1, 99, 100, 150, 200):
        tds.adjust_learning_rate(opts, 0, cfg, epoch=ep)
        table.append([ep] + [o.param_groups[0]["lr"] for o in opts])
    save("g12_lr", table=np.array(table, dtype=np.float64))


if __name__ == "__main__":
    main()
