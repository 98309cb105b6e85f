"""Pin the CPU oracle (oracle/fovealseg_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fovealseg  # noqa: F401
import fovealseg_oracle as O
from fovealseg.weights import apply_name_keyed_init

torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def oracle():
    m = O.OracleDeformSeg()
    apply_name_keyed_init(m)
    return m


def reinit(m):
    apply_name_keyed_init(m)
    return m


def test_g1_constants(golden):
    g = golden("g1_constants")
    assert np.array_equal(O.make_gaussian(91, 45), g["gaussian"])
    m = O.OracleDeformSeg()
    assert np.array_equal(m.filter.weight[0, 0].detach().numpy(), g["filter"])
    assert np.array_equal(m.P_basis.numpy(), g["p_basis"])
    # separability of the Gaussian
    g1 = O.gaussian_1d(91, 45)
    assert np.abs(np.outer(g1, g1) - g["gaussian"]).max() < 2e-16


def _synth(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(B, 3, H, W, generator=g)
    Fp = torch.rand(B, 2, generator=g) * 0.8 + 0.1
    return X, Fp


@pytest.mark.parametrize("H", [128, 640])
def test_g2_lowres(golden, H):
    g = golden(f"g2_lowres_{H}")
    X, Fp = _synth(2, H, H, int(g["seed"]))
    assert np.array_equal(Fp.numpy(), g["focus"])
    out = O.lowres_input(X, Fp, 80, 80)
    assert np.abs(out.numpy() - g["x_low"]).max() <= 1e-6


def test_g3_saliency(golden, oracle):
    g = golden("g3_saliency")
    x_low = T(g["x_low"])
    for mode in ("eval", "train"):
        reinit(oracle).train(mode == "train")
        with torch.no_grad():
            s = oracle.net_compress(oracle.localization(x_low))
            xs = F.softmax(s.view(3, -1), 1).view(3, 1, 80, 80)
        assert np.abs(s.numpy() - g["logit_" + mode]).max() <= 1e-5
        assert np.abs(xs.numpy() - g["xs_" + mode]).max() <= 1e-7
    reinit(oracle)


def test_g4_grid(golden, oracle):
    g = golden("g4_grid")
    xs = T(g["xs"]).clone().requires_grad_(True)
    grid = oracle.grid_from_saliency(xs)
    assert np.abs(grid.detach().numpy() - g["grid"]).max() <= 1e-6
    assert np.array_equal(g["grid"], g["grid_y"])
    (grid * T(g["cot"])).sum().backward()
    ref = g["dxs"]
    assert np.abs(xs.grad.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    # the reference's own fp32 result vs an fp64 evaluation (SURVEY §7 budget: <=3e-5)
    g64 = O.create_grid_f64(T(g["xs"]), 45)
    assert np.abs(g64.numpy() - g["grid"]).max() <= 3e-5


@pytest.mark.parametrize("tag", ["128x128", "200x136"])
def test_g5_grid_sample(golden, tag):
    g = golden("g5_gridsample_" + tag)
    grid = T(g["grid"]).clone().requires_grad_(True)
    xs = F.grid_sample(T(g["x"]), grid, align_corners=False)
    ys = F.grid_sample(T(g["y"]), grid, align_corners=False).squeeze(1)
    assert np.array_equal(xs.detach().numpy(), g["x_sampled"])
    assert np.array_equal(ys.detach().long().numpy(), g["label"])
    (xs * T(g["cot"])).sum().backward()
    assert np.abs(grid.grad.numpy() - g["dgrid"]).max() <= 1e-6 * max(1.0, np.abs(g["dgrid"]).max())


@pytest.mark.parametrize("H", [128, 640])
def test_g6_inverse_maps(golden, H):
    g = golden(f"g6_inverse_{H}")
    u, v, nan = O.inverse_index_maps(T(g["grid"]), H, H)
    assert np.array_equal(u.numpy(), g["u"]) and np.array_equal(v.numpy(), g["v"])
    assert np.array_equal(nan.sum(dim=(1, 2)).numpy(), g["nan_count"])
    assert np.array_equal(np.packbits(nan.numpy()), g["nanmask"])


def _sub(m, path):
    for p in path.split("."):
        m = m[int(p)] if p.isdigit() else getattr(m, p)
    return m


@pytest.mark.parametrize("name", ["basic", "bottleneck", "hrmodule4"])
@pytest.mark.parametrize("mode", ["eval", "train_p0"])
def test_g7_blocks(golden, oracle, name, mode):
    g = golden(f"g7_{name}_{mode}")
    prefix = str(g["prefix"])
    reinit(oracle)
    blk = _sub(oracle, prefix)
    blk.train(mode != "eval")
    n_in = sum(1 for k in g.files if k.startswith("in"))
    ins = [T(g[f"in{i}"]).clone().requires_grad_(True) for i in range(n_in)]
    ctx = O._Ctx(mode != "eval", (lambda n, t: t) if mode == "train_p0" else None)
    rel = prefix.split(".", 1)[1]
    if name == "basic":
        outs = [blk(ins[0], ctx, rel)]
    elif name == "bottleneck":
        outs = [blk(ins[0])]
    else:
        outs = blk(ins, ctx, rel)
    blk.zero_grad()
    sum((o * T(g[f"cot{i}"])).sum() for i, o in enumerate(outs)).backward()
    for i, o in enumerate(outs):
        ref = g[f"out{i}"]
        assert np.abs(o.detach().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    for i, t in enumerate(ins):
        ref = g[f"din{i}"]
        assert np.abs(t.grad.numpy() - ref).max() <= 1e-4 * max(1e-3, np.abs(ref).max())
    params = dict(blk.named_parameters())
    for k in g.files:
        if k.startswith("dw:") or k.startswith("dgamma:"):
            gr = params[k.split(":", 1)[1]].grad
            ref = g[k]
            if gr.shape != ref.shape:
                gr = gr[:16, :16]
            assert np.abs(gr.numpy() - ref).max() <= 2e-4 * max(1e-3, np.abs(ref).max()), k
    reinit(oracle)


def test_g8_hrnet_eval(golden, oracle):
    g = golden("g8_hrnet_eval")
    reinit(oracle).eval()
    with torch.no_grad():
        feat = oracle.encoder(T(g["x"]), return_feature_maps=True)[0]
    scale = np.abs(g["crop"]).max()
    assert np.abs(feat[0, :, 32:48, 32:48].numpy() - g["crop"]).max() <= 1e-4 * max(1.0, scale)
    assert np.abs(feat.mean(dim=(0, 2, 3)).numpy() - g["chan_mean"]).max() <= 1e-4 * max(1.0, scale)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_g9_c1(golden, oracle, mode):
    g = golden(f"g9_c1_{mode}")
    reinit(oracle)
    oracle.decoder.train(mode == "train")
    gg = torch.Generator().manual_seed(int(g["seed"]))
    f9 = (torch.randn(2, 960, 80, 80, generator=gg) * 0.5).requires_grad_(True)
    pred = oracle.decoder([f9])
    cot = torch.randn(pred.shape, generator=gg) * 0.01
    oracle.decoder.zero_grad()
    (pred * cot).sum().backward()
    assert np.abs(pred[:, :50, 0, 0].detach().numpy() - g["pred_ch0"]).max() <= 1e-5
    assert np.abs(pred[:, 50].detach().numpy() - g["pred_last"]).max() <= 1e-5
    ref = g["dfeat_crop"]
    assert np.abs(f9.grad[:, ::60, 20:36, 20:36].numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    ref = g["dfc"]
    assert np.abs(oracle.decoder.cls_net.fc.weight.grad.numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    reinit(oracle)


def test_g10_losses(golden):
    g = golden("g10_losses")
    pred = T(g["pred"]).clone().requires_grad_(True)
    gt = T(g["gt"])
    fl = O.focal_loss(pred, gt)
    dl = O.dice_loss_multiclass(pred, gt)
    (fl + dl).backward()
    assert abs(float(fl) - float(g["focal"])) <= 1e-6
    assert abs(float(dl) - float(g["dice"])) <= 1e-6
    assert np.abs(pred.grad.numpy() - g["dpred"]).max() <= 1e-6 * max(1.0, np.abs(g["dpred"]).max())
    accs = O.accuracies(pred.detach(), gt)
    assert np.abs(np.array([float(a) for a in accs]) - g["accs"]).max() <= 1e-6
    xs = T(g["xs"]).clone().requires_grad_(True)
    ii = torch.arange(256, dtype=torch.float32)
    # regenerate the disc masks exactly as make_goldens.synth_batch does
    gg = torch.Generator().manual_seed(int(g["y_seed"]))
    torch.rand(3, 3, 256, 256, generator=gg)
    Fp = torch.rand(3, 2, generator=gg) * 0.8 + 0.1
    cy = (Fp[:, 0] * 255)[:, None, None]
    cx = (Fp[:, 1] * 255)[:, None, None]
    Y = (((ii[None, :, None] - cy) ** 2 + (ii[None, None, :] - cx) ** 2) <= (0.15 * 256) ** 2).float().unsqueeze(1)
    assert np.abs(F.interpolate(Y, size=(80, 80), mode="area").numpy() - g["area"]).max() <= 1e-6
    el = O.edge_loss(xs, Y, 80, 80, 100.0)
    el.backward()
    assert abs(float(el) - float(g["edge"])) <= 1e-6
    assert np.abs(xs.grad.numpy() - g["dxs"]).max() <= 1e-5 * np.abs(g["dxs"]).max()


def test_g11_end_to_end(golden, oracle):
    g = golden("g11_e2e_eval")
    reinit(oracle).eval()
    feed = {"img_data": T(g["x"]), "seg_label": T(g["y"]).clone(), "focus_point": T(g["focus"]), "cls_label": T(g["cls"])}
    with torch.no_grad():
        outs = oracle(feed, is_inference=True)
    got = np.array([float(o) for o in outs])
    assert np.array_equal(feed["seg_label"].numpy(), g["label"])
    assert np.abs(got - g["outs"]).max() <= 1e-4, (got, g["outs"])

    g = golden("g11_e2e_train_p0")
    reinit(oracle).train()
    feed = {"img_data": T(g["x"]), "seg_label": T(g["y"]).clone(), "focus_point": T(g["focus"]), "cls_label": T(g["cls"])}
    oracle.zero_grad()
    loss, acc, edge = oracle(feed, drop_fn=lambda n, t: t)
    loss.backward()
    got = np.array([float(loss), float(acc), float(edge)])
    assert np.abs(got - g["outs"]).max() <= 1e-4, (got, g["outs"])
    params = dict(oracle.named_parameters())
    for n, ref in zip(g["gn_names"], g["gn"]):
        gn = float(params[str(n)].grad.norm())
        assert abs(gn - float(ref)) <= 1e-3 * max(abs(float(ref)), 1e-6), (n, gn, ref)
    reinit(oracle)


def test_g12_lr(golden):
    g = golden("g12_lr")
    for row in g["table"]:
        ep = int(row[0])
        for lr in row[1:]:
            assert lr == O.lr_for_epoch(ep)


def test_g13_segformer_oracle_vs_transformers(golden):
    """oracle/segformer_oracle.py against transformers 5.15.0 SegformerModel (tests/golden/make_segformer_golden.py)."""
    import segformer_oracle as SO
    g = golden("g13_segformer")
    o = SO.OracleSegformer()
    apply_name_keyed_init(o)
    o.eval()
    with torch.no_grad():
        outs = o.stages(T(g["x"]))
        cat = o(T(g["x"]))[0]
    assert cat.shape == (1, 1024, 80, 80)
    assert np.abs(outs[3][0].numpy() - g["stage3"]).max() <= 1e-4
    assert np.abs(cat[0, :, 32:48, 32:48].numpy() - g["crop"]).max() <= 1e-4
    assert np.abs(cat.mean(dim=(0, 2, 3)).numpy() - g["chan_mean"]).max() <= 1e-4


def test_dice_known_answers_unpinned_toolbelt():
    """A18: pytorch_toolbelt 0.8.0 is absent, so the Dice restatement cannot be pinned to the package; these are hand-derived
    known answers (tests/golden/kat_dice_unpinned.json, derivation in each case's "why") that any correct restatement of
    DiceLoss('multiclass', from_logits=True, smooth=0, eps=1e-7) must reproduce: sums over batch and pixels, classes absent
    from the target zeroed, mean over ALL classes."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_dice_unpinned.json")) as f:
        kat = json.load(f)
    for c in kat["cases"]:
        pred = torch.tensor(c["logits"], dtype=torch.float32)
        gt = torch.tensor(c["gt"], dtype=torch.int64)
        got = float(O.dice_loss_multiclass(pred, gt))
        assert abs(got - c["dice"]) <= 2e-7, (c["name"], got, c["dice"])
