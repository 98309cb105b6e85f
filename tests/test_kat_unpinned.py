"""Hand-derived known answers for the pieces whose third-party source is absent (VERDICT r1 "missing" #2): SegFormer
(transformers 4.46.2) and DeepLab (torchvision 0.19.1) are PARITY UNPINNED -- the reference holds no fixture at their boundary and
neither package can be imported at the pinned version.  What can be checked without them: inputs / weights crafted so that the
published architecture's output has a closed form, evaluated here in fp64 numpy independently of both the oracle and the HIP
modules.  CPU tests check the oracles, `-m gpu` tests check the HIP modules, against the SAME closed forms.
(The Dice loss known answers, A18, live in tests/golden/kat_dice_unpinned.json.)
"""
import numpy as np
import pytest
import torch

import fovealseg  # noqa: F401
import segformer_oracle as SO
import deeplab_oracle as DO

LN_EPS = 1e-6


# ---------------------------------------------------------------------------------------------------------------
# closed forms
# ---------------------------------------------------------------------------------------------------------------
def ln64(v, eps=LN_EPS):
    v = np.asarray(v, dtype=np.float64)
    mu = v.mean(-1, keepdims=True)
    var = ((v - mu) ** 2).mean(-1, keepdims=True)
    return (v - mu) / np.sqrt(var + eps)


def crafted_segformer_layer(layer, b_o, b_2):
    """Zero every weight matrix of a Mix-Transformer block, keep LayerNorms at (1, 0): attention output = its dense bias b_o,
    Mix-FFN output = dense2's bias b_2, so the block is x -> x + b_o + b_2 (two residual adds, eval mode)."""
    with torch.no_grad():
        for p in layer.parameters():
            p.zero_()
        for name, p in layer.named_parameters():
            if "layer_norm" in name and name.endswith("weight"):
                p.fill_(1.0)
        layer.attention.output.dense.bias.copy_(torch.as_tensor(b_o, dtype=torch.float32))
        layer.mlp.dense2.bias.copy_(torch.as_tensor(b_2, dtype=torch.float32))


def attention_cases():
    rng = np.random.default_rng(3)
    # (1) identical keys: softmax is uniform whatever q is -> output = mean of the value rows
    q = rng.standard_normal((1, 40, 64))
    k = np.tile(rng.standard_normal((1, 1, 64)), (1, 7, 1))
    v = rng.standard_normal((1, 7, 64))
    yield "uniform", q, k, v, np.tile(v.mean(1, keepdims=True), (1, 40, 1))
    # (2) one dominant key: q.k_j / 8 = 100 for j = 2, 0 elsewhere -> weights (e^100, 1, 1, ...) / Z -> v_2 to 1e-40
    q = np.zeros((1, 33, 64)); q[..., 0] = 1.0
    k = np.zeros((1, 5, 64)); k[0, 2, 0] = 800.0
    v = rng.standard_normal((1, 5, 64))
    yield "dominant", q, k, v, np.tile(v[:, 2:3], (1, 33, 1))
    # (3) two keys with logits (0, ln 3): weights (1/4, 3/4)
    q = np.zeros((1, 3, 64)); q[..., 1] = 8.0
    k = np.zeros((1, 2, 64)); k[0, 1, 1] = np.log(3.0)
    v = np.zeros((1, 2, 64)); v[0, 0, :] = 4.0; v[0, 1, :] = -4.0
    yield "quarter", q, k, v, np.full((1, 3, 64), 0.25 * 4.0 + 0.75 * -4.0)


# ---------------------------------------------------------------------------------------------------------------
# CPU: the oracles
# ---------------------------------------------------------------------------------------------------------------
def test_segformer_oracle_known_answers_unpinned():
    o = SO.OracleSegformer().eval()
    layer = o.segformer.encoder.block[0][1]                 # hidden 64, 1 head, sequence reduction 8
    b_o, b_2 = np.linspace(-1, 1, 64), np.linspace(0.5, -0.25, 64)
    crafted_segformer_layer(layer, b_o, b_2)
    x = torch.randn(2, 16 * 16, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = layer(x, 16, 16, SO.Hooks(), "p", False)
    assert np.abs(y.numpy() - (x.numpy().astype(np.float64) + b_o + b_2)).max() <= 1e-6
    for name, q, k, v, want in attention_cases():           # the oracle's attention arithmetic (1 head, d = 64)
        probs = torch.softmax(torch.from_numpy(q).float() @ torch.from_numpy(k).float().transpose(-1, -2) / 8.0, -1)
        got = (probs @ torch.from_numpy(v).float()).numpy()
        assert np.abs(got - want).max() <= 2e-6, name
    ln = torch.nn.LayerNorm(4, eps=LN_EPS)
    got = ln(torch.tensor([[1.0, 2.0, 3.0, 4.0]])).detach().numpy()
    assert np.abs(got - (np.array([1, 2, 3, 4.0]) - 2.5) / np.sqrt(1.25 + LN_EPS)).max() <= 1e-6


def _delta_dilated_case():
    """3x3 conv, dilation 2, padding 2, one input channel, on a unit impulse at (5,5) of a 11x11 map: the output is the kernel laid
    out at the nine positions (5 - 2(r-1), 5 - 2(s-1)) -- cross-correlation -- and zero elsewhere."""
    w = np.arange(1.0, 10.0).reshape(3, 3)
    x = np.zeros((11, 11)); x[5, 5] = 1.0
    want = np.zeros((11, 11))
    for r in range(3):
        for s in range(3):
            want[5 - 2 * (r - 1), 5 - 2 * (s - 1)] = w[r, s]
    return x, w, want


def test_deeplab_oracle_known_answers_unpinned():
    x, w, want = _delta_dilated_case()
    conv = torch.nn.Conv2d(1, 1, 3, 1, 2, 2, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w).float().view(1, 1, 3, 3))
        got = conv(torch.from_numpy(x).float().view(1, 1, 11, 11))[0, 0].numpy()
    assert np.abs(got - want).max() == 0.0
    # ASPP image-pooling branch of the oracle: AdaptiveAvgPool2d(1) -> 1x1 conv -> BN (eval: mean 0, var 1) -> ReLU -> broadcast
    o = DO.OracleDeepLab().eval()
    aspp = o.deeplab.classifier[0]
    pool = aspp.convs[4]
    with torch.no_grad():
        pool[1].weight.zero_()
        for c in range(256):
            pool[1].weight[c, c, 0, 0] = 1.0 if c % 2 == 0 else -1.0          # channel c -> +-mean of channel c
        bn = pool[2]
        bn.weight.fill_(1.0); bn.bias.zero_(); bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        f = torch.rand(2, 2048, 10, 10, generator=torch.Generator().manual_seed(2))
        got = pool(f)                                                          # (2,256,1,1)
    m = f.double().mean((2, 3)).numpy()[:, :256]
    sign = np.where(np.arange(256) % 2 == 0, 1.0, -1.0)
    want = np.maximum(sign * m / np.sqrt(1.0 + 1e-5), 0.0)
    assert np.abs(got[:, :, 0, 0].numpy() - want).max() <= 1e-6
    # output stride: 80x80 input -> /2 (stem) /2 (maxpool) /2 (layer2), layers 3-4 dilated -> 10x10, up-sampled back to 80x80
    with torch.no_grad():
        feat = o.deeplab.backbone(torch.rand(1, 3, 80, 80))
    assert tuple(feat.shape) == (1, 2048, 10, 10)


def test_deeplab_size_note_of_the_reference_unpinned():
    """The one datum the reference itself holds about its DeepLab encoder: the note its author left behind a (commented-out) thop profile of
    `CustomDeepLab` -- `Params: {params / 1e6:.4f} M` ... `#resnet101 11G 58M` (models/deeplab.py:436-439).  torchvision 0.19.1's
    deeplabv3_resnet101 with the reference's three replacements (models/deeplab.py:11-49) adds up to 58 660 352 parameters from its published
    layer shapes: ResNet-101 without the fc 42 500 160; ASPP(2048, (12, 24, 36)) 15 535 104; classifier[1] Conv1x1(256, 512) + BN(512) and
    classifier[4] Conv1x1(512, 960), both with bias, 625 088.  A weak pin (a count, not arithmetic) -- the encoder stays PARITY UNPINNED --
    but the oracle and the HIP-side module must both land on it, tensor by tensor alike."""
    from fovealseg import deeplab as D
    o, m = DO.OracleDeepLab(), D.deeplab()
    no = sum(p.numel() for p in o.parameters())
    nm = sum(p.numel() for p in m.parameters())
    assert no == nm == 58_660_352
    assert int(nm / 1e6) == 58                                                 # "58M"
    parts = {"backbone": 42_500_160, "classifier.0": 15_535_104}
    for net in (o, m):
        for prefix, want in parts.items():
            got = sum(p.numel() for k, p in net.named_parameters() if k.startswith("deeplab." + prefix + "."))
            assert got == want, (prefix, got, want)
    shapes_o = {k: tuple(p.shape) for k, p in o.named_parameters()}
    shapes_m = {k: tuple(p.shape) for k, p in m.named_parameters()}
    assert shapes_o == shapes_m


# ---------------------------------------------------------------------------------------------------------------
# GPU: the HIP modules against the same closed forms
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_segformer_hip_known_answers_unpinned():
    from fovealseg import ops, segformer as S
    m = S.segformer().to("cuda").eval()
    layer = m.segformer.encoder.block[0][1]
    b_o, b_2 = np.linspace(-1, 1, 64), np.linspace(0.5, -0.25, 64)
    crafted_segformer_layer(layer, b_o, b_2)
    x = torch.randn(2, 16 * 16, 64, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        y = layer(x.view(2, 16, 16, 64).cuda()).cpu().view(2, 256, 64)
    assert np.abs(y.numpy() - (x.numpy().astype(np.float64) + b_o + b_2)).max() <= 1e-6
    for name, q, k, v, want in attention_cases():
        got = ops.Attention.apply(torch.from_numpy(q).float().cuda(), torch.from_numpy(k).float().cuda(), torch.from_numpy(v).float().cuda(),
                                  1, 0.0, 0).cpu().numpy()
        assert np.abs(got - want).max() <= 2e-6, name
    g, b = torch.ones(4, device="cuda"), torch.zeros(4, device="cuda")
    got = ops.LayerNorm.apply(torch.tensor([[1.0, 2.0, 3.0, 4.0]], device="cuda"), g, b, LN_EPS).cpu().numpy()
    assert np.abs(got - (np.array([1, 2, 3, 4.0]) - 2.5) / np.sqrt(1.25 + LN_EPS)).max() <= 1e-6


@pytest.mark.gpu
def test_deeplab_hip_known_answers_unpinned():
    from fovealseg import ops, deeplab as D
    x, w, want = _delta_dilated_case()
    # the dilated conv through the C ABI: 4 channels (the aligned kernels' minimum), the impulse and the kernel in channel pair (1 -> 2)
    xd = torch.zeros(1, 11, 11, 4, device="cuda"); xd[0, :, :, 1] = torch.from_numpy(x).float().cuda()
    wl = torch.zeros(4, 4, 3, 3); wl[2, 1] = torch.from_numpy(w).float()
    wd = ops.new_rsck_weight(4, 4, 3, 3, device="cuda"); wd.copy_(wl)
    for mode in ("f32", "bf16x3", "f16x2"):
        fovealseg.hip.set_conv_precision(mode)
        try:
            got = ops.conv2d_fwd(xd, wd, None, 1, 2, dil=2)[0, :, :, 2].cpu().numpy()
        finally:
            fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())
        assert np.abs(got - want).max() == 0.0, mode                # small integers: exact in every mode
    m = D.deeplab().to("cuda").eval()
    pool = m.deeplab.classifier[0].convs[4]
    with torch.no_grad():
        wgt = torch.zeros(256, 2048, 1, 1)
        for c in range(256):
            wgt[c, c, 0, 0] = 1.0 if c % 2 == 0 else -1.0
        pool[1].weight.copy_(wgt.cuda())
        bn = pool[2]
        bn.weight.fill_(1.0); bn.bias.zero_(); bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        f = torch.rand(2, 2048, 10, 10, generator=torch.Generator().manual_seed(2))
        got = pool(f.permute(0, 2, 3, 1).contiguous().cuda()).cpu()            # (2,1,1,256) NHWC
        feat = m.deeplab.backbone(torch.rand(1, 80, 80, 3, device="cuda"))
    mm = f.double().mean((2, 3)).numpy()[:, :256]
    sign = np.where(np.arange(256) % 2 == 0, 1.0, -1.0)
    assert np.abs(got.view(2, 256).numpy() - np.maximum(sign * mm / np.sqrt(1.0 + 1e-5), 0.0)).max() <= 1e-6
    assert tuple(feat.shape) == (1, 10, 10, 2048)
