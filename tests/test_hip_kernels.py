"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle / torch-CPU fp32 and
the committed golden fixtures.  Run on the MI355X box:  python -m pytest tests -m gpu -x -q
Tolerances: bit-exact for integer/index work and grid_sample forward; fp32 otherwise, stated per test.
"""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import fovealseg  # noqa: E402
from fovealseg import ops  # noqa: E402
import fovealseg_oracle as O  # noqa: E402

DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def nhwc(x):          # (B,C,H,W) cpu -> (B,H,W,C) device contiguous
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(x):          # (B,H,W,C) device -> (B,C,H,W) cpu
    return x.detach().permute(0, 3, 1, 2).contiguous().cpu()


def rsck_param(w):    # logical (Co,Ci,R,S) cpu -> device tensor with RSCK storage
    p = ops.new_rsck_weight(*w.shape, device=DEV)
    p.copy_(w)
    return p


def relerr(a, b):
    """max |a-b| / max |b| over ALL elements (no quantile, no per-mode allowance: the comparisons below are made
    well-conditioned instead -- the branch every activation took on the device is replayed in the CPU oracle, the way
    Dropout masks are, so both sides differentiate the same smooth function; see `replay`)."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rmsrel(a, b):
    """||a-b||_2 / ||b||_2 (whole-tensor relative error of a gradient tensor)."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def analytic_zero_grad(name, training):
    """Parameters whose gradient is analytically zero, so both sides hold only rounding noise: a conv bias in front of a
    batch-statistics BatchNorm (C1's cls_net convs carry biases, model_utils.py:227-236), and the bias of the 1x1 conv in
    front of the spatial softmax (shift invariance)."""
    if name.endswith("net_compress.conv_last.bias"):
        return True
    return training and "cls_net.layer" in name and name.endswith(".0.bias")


class replay:
    """with replay(root, trace): the oracle's activation sites take the branches recorded in `trace` (an ops.ACT_TRACE
    list filled by a forward of the HIP module tree `root`; names are module paths relative to `root`)."""

    def __init__(self, root, trace, prefix=""):
        names = {id(m): n for n, m in root.named_modules()}
        self.lo, self.hi = {}, {}
        for key, act, z in trace:
            if isinstance(key, tuple):
                name = (names[id(key[0])] + f".fuse{key[1]}").strip(".")
            else:
                name = names[id(key)]
            name = (prefix + "." + name).strip(".")
            zd = z.detach()                                     # masks are formed on the device: 1 byte per element crosses PCIe
            if act == 2:
                self.lo[name], self.hi[name] = nchw((zd > 0) & (zd < 6)), nchw(zd >= 6)
            else:
                self.lo[name] = nchw(zd > 0)

    def __enter__(self):
        O.ACT_REPLAY, O.ACT_REPLAY_HI = self.lo, self.hi
        return self

    def __exit__(self, *exc):
        O.ACT_REPLAY = O.ACT_REPLAY_HI = None


class traced:
    """with traced() as tr: HIP forwards append their activation outputs to tr."""

    def __enter__(self):
        ops.ACT_TRACE = []
        return ops.ACT_TRACE

    def __exit__(self, *exc):
        ops.ACT_TRACE = None


@pytest.fixture(params=["f32", "bf16x3", "f16x2"])
def prec(request):
    fovealseg.hip.set_conv_precision(request.param)
    yield request.param
    fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


# ------------------------------------------------------------------------------------------------
# convolution engine
# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, bias
    (2, 20, 20, 64, 64, 3, 1, False),
    (3, 17, 13, 3, 64, 3, 1, False),       # stem, scalar path, ragged M
    (2, 16, 16, 5, 192, 3, 1, False),      # saliency first layer
    (2, 16, 16, 192, 24, 3, 1, False),     # Cout < tile
    (2, 20, 20, 64, 128, 3, 2, False),     # stride 2
    (2, 20, 20, 256, 64, 1, 1, False),     # 1x1
    (1, 12, 12, 960, 240, 3, 1, False),    # C1 cbr
    (2, 20, 20, 96, 512, 3, 4, True),      # cls_net stride 4 + bias
    (2, 20, 20, 96, 64, 1, 4, True),       # 1x1 stride 4 downsample
    (2, 19, 22, 32, 48, 3, 4, False),      # stride > filter reach, ragged size: empty bwd-data classes are zero-filled
    (2, 17, 17, 32, 32, 3, 3, False),      # stride = filter size
    (1, 13, 9, 16, 32, 1, 3, False),
    (5, 1, 1, 512, 51, 1, 1, True),        # FC as 1x1 conv, scalar path
    (1, 10, 10, 512, 512, 3, 1, False),
    (2, 9, 11, 64, 96, 3, 1, True),        # odd width: the direct halo kernel (the F(2,3) row kernel needs pixel pairs), + bias
    (2, 10, 10, 64, 64, 3, 1, False, 2),    # dilation 2 (DeepLab layer3)
    (1, 10, 10, 128, 64, 3, 1, False, 12),  # ASPP rate 12: only the centre tap is in range
    (2, 23, 17, 3, 64, 7, 2, False),        # ResNet stem 7x7 stride 2 (generic kernel)
    (3, 21, 19, 48, 64, 3, 2, False),       # one channel tile, stride 2: bwd-weight with all nine taps in one launch (parity planes), ragged
    (2, 40, 40, 64, 64, 3, 2, True),        # the HRNet fuse down-path shape of that kernel
    (2, 20, 20, 128, 96, 3, 4, True),       # stride >= filter with 64-aligned input channels: forward = the 1x1 GEMM kernel over gathered rows
    (3, 19, 22, 64, 512, 3, 4, False),      # ... ragged size, 128-column workgroups
    (2, 16, 16, 64, 64, 1, 4, True),        # ... 1x1 stride 4
    (2, 17, 17, 192, 32, 3, 3, False),      # ... stride = filter size, three rounds per tap
    (2, 18, 21, 96, 128, 3, 3, False),      # bwd-data of such layers with 64-aligned OUTPUT channels: one GEMM per tap, rows scattered to dX
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, prec):
    B, H, W, Ci, Co, k, s, has_bias = case[:8]
    dil = case[8] if len(case) > 8 else 1
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) if has_bias else None
    pad = dil * (k // 2)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, s, pad, dil)
    cot = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(cot)

    xd, wd = nhwc(x), rsck_param(w)
    bd = b.to(DEV) if has_bias else None
    y = ops.conv2d_fwd(xd, wd, bd, s, pad, dil=dil)
    assert relerr(nchw(y), y_ref.detach()) <= 2e-5
    dyd = nhwc(cot)
    dx = ops.conv2d_bwd_data(dyd, wd, xd.shape, s, pad, dil)
    assert relerr(nchw(dx), xr.grad) <= 2e-5
    dw = ops.conv2d_bwd_weight(xd, dyd, w.shape, s, pad, dil=dil)
    assert dw.shape == w.shape
    assert relerr(dw.cpu(), wr.grad) <= 5e-5


# Deterministic mode (include/fovealseg.h fs_set_deterministic; the reference's cudnn.deterministic = True, train_deform_semantic.py:680-681):
# every bwd-weight kernel family -- 3x3 class kernel, its stride-2 / stride-4 tap classes, the transform-domain kernel (long pixel loops),
# the linear GEMM (+ fused bias sums), the fp32 tap kernels and the generic unaligned kernel -- writes per-split partial tiles and sums
# them in index order: two calls are bit-identical, equal the atomics' result to rounding, accumulate on top of an existing gradient,
# and a missing scratch is an error, never a silent fall-back to the atomics.
DET_WGRAD_CASES = [
    (8, 40, 40, 64, 64, 3, 1), (4, 40, 40, 64, 128, 3, 2), (4, 40, 40, 64, 64, 3, 2), (2, 40, 40, 96, 64, 3, 4), (4, 80, 80, 192, 24, 3, 1),
    (4, 40, 40, 64, 256, 1, 1), (4, 20, 20, 256, 64, 1, 1), (2, 16, 16, 64, 64, 1, 2), (3, 1, 1, 512, 51, 1, 1), (2, 23, 17, 3, 64, 7, 2),
]


@pytest.mark.parametrize("case", DET_WGRAD_CASES)
def test_bwd_weight_deterministic_mode(case, prec):
    B, H, W, Ci, Co, k, s = case
    H_ = fovealseg.hip
    g = torch.Generator().manual_seed(Ci * 7 + Co + k)
    pad = k // 2
    x = torch.randn(B, H, W, Ci, generator=g).to(DEV)
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    dy = torch.randn(B, Ho, Wo, Co, generator=g).to(DEV)
    shape = (Co, Ci, k, k)
    ref = ops.conv2d_bwd_weight(x, dy, shape, s, pad).clone()                 # atomics
    assert not H_.get_deterministic()
    H_.set_deterministic(True)
    try:
        assert int(H_.load().fs_conv2d_bwd_weight_ws_bytes(Ci, Co, k, k, s, pad, 1)) > 0
        a = ops.conv2d_bwd_weight(x, dy, shape, s, pad).clone()
        b = ops.conv2d_bwd_weight(x, dy, shape, s, pad).clone()
        assert torch.equal(a, b)
        assert relerr(a, ref) <= 2e-6
        base = torch.randn(k, k, Ci, Co, generator=g).to(DEV).permute(3, 2, 0, 1)       # accumulate: the ordered sum starts from dw
        acc = base.clone(memory_format=torch.preserve_format)
        ops.conv2d_bwd_weight(x, dy, shape, s, pad, out=acc, accumulate=True)
        assert relerr(acc - base, a) <= 1e-5
        dw = torch.empty(k, k, Ci, Co, device=DEV)
        with pytest.raises(H_.HipLibraryError):                                            # no scratch in deterministic mode: loud
            H_.call("fs_conv2d_bwd_weight", H_.ptr(x), H_.ptr(dy), H_.ptr(dw), B, H, W, Ci, Ho, Wo, Co, k, k, s, pad, 1, 0, None, 0)
        if k == 1 and s == 1 and H_.linear_bwd_weight_bias_ok(B * H * W, Ci, Co):
            outs = []
            for _ in range(2):
                dw1, db1 = torch.empty(Ci, Co, device=DEV), torch.empty(Co, device=DEV)
                nb = int(H_.load().fs_linear_bwd_weight_bias_ws_bytes(Ci, Co))
                ws = torch.empty(nb, device=DEV, dtype=torch.uint8)
                H_.call("fs_linear_bwd_weight_bias", H_.ptr(x), H_.ptr(dy), H_.ptr(dw1), H_.ptr(db1), B * H * W, Ci, Co, 0, 0, H_.ptr(ws), nb)
                outs.append((dw1, db1))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
            assert relerr(outs[0][0], ref.permute(2, 3, 1, 0)[0, 0]) <= 2e-6
            assert relerr(outs[0][1], dy.reshape(-1, Co).double().sum(0).float()) <= 2e-6
    finally:
        H_.set_deterministic(False)
    if not (k == 3 and s in (2, 3) and prec != "f32"):          # (strided 3x3 layers take the store + ordered-reduce route in every mode)
        assert int(H_.load().fs_conv2d_bwd_weight_ws_bytes(Ci, Co, k, k, s, pad, 1)) == 0


# Weight packs that outlive the call (include/fovealseg.h: fs_conv2d_pack / fs_conv2d_ws_mode; ops._pack_for, ops.repack_weights).
PACK_CASES = [
    (4, 16, 16, 64, 64, 3, 1),       # F(2,3) kernel (even width)
    (2, 15, 15, 64, 64, 3, 1),       # halo-tiled kernel (odd width)
    (2, 20, 20, 256, 256, 3, 1),     # eight-wave F(2,3) form in bf16x3
    (4, 16, 16, 64, 128, 3, 2),      # stride 2: parity-plane forward, four-parity bwd-data
    (4, 16, 16, 64, 256, 1, 1),      # 1x1 GEMM kernel
    (2, 16, 16, 32, 64, 5, 1),       # tap-class kernel (5x5)
    (2, 16, 16, 64, 64, 3, 3),       # stride 3: bwd-data re-packs per parity class -> no persistent pack
]


@pytest.mark.parametrize("case", PACK_CASES)
def test_persistent_weight_pack(case, prec):
    """fs_conv2d_pack once, then run-only calls: bit-identical to the pack-then-run call; the run-only call really reads the scratch
    (new weights, old pack -> old result); a problem without a persistent pack refuses the run-only mode."""
    B, H, W, Ci, Co, k, s = case
    H_ = fovealseg.hip
    lib = H_.load()
    g = torch.Generator().manual_seed(Ci + 3 * Co + k + s)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, Ci, generator=g).to(DEV)
    dy = torch.randn(B, Ho, Wo, Co, generator=g).to(DEV)
    w = (torch.randn(k, k, Ci, Co, generator=g) / (k * Ci ** 0.5)).to(DEV)
    w2 = w * 1.5
    shape = (B, H, W, Ci, Ho, Wo, Co, k, k, s, pad, 1)
    for transposed in (0, 1):
        n = H_.conv_workspace_bytes(*shape[1:], transposed)
        if prec == "f32":
            assert n == 0
            continue
        src, dst = (dy, torch.empty(B, H, W, Ci, device=DEV)) if transposed else (x, torch.empty(B, Ho, Wo, Co, device=DEV))

        def run(weights, ws, packed):
            out = torch.empty_like(dst)
            if transposed:
                args = ("fs_conv2d_bwd_data", H_.ptr(src), H_.ptr(weights), H_.ptr(out), *shape, H_.ptr(ws), n, None)
            else:
                args = ("fs_conv2d_fwd", H_.ptr(src), H_.ptr(weights), None, H_.ptr(out), *shape, 0.0, 0, H_.ptr(ws), n, None)
            (H_.call_packed if packed else H_.call)(*args)
            return out

        if n == 0:
            continue
        ref = run(w, torch.empty(n, device=DEV, dtype=torch.uint8), False)
        persistent = int(lib.fs_conv2d_pack_persistent(*shape, transposed, n))
        choice = int(lib.fs_conv2d_kernel_choice(*shape, transposed, n))
        if not persistent:
            assert choice in (0, 1, 3), choice
            with pytest.raises(H_.HipLibraryError):
                run(w, torch.empty(n, device=DEV, dtype=torch.uint8), True)
            with pytest.raises(H_.HipLibraryError):
                H_.call("fs_conv2d_pack", H_.ptr(w), *shape, transposed, H_.ptr(torch.empty(n, device=DEV, dtype=torch.uint8)), n, None)
            assert int(lib.fs_conv2d_ws_mode(0)) == 0          # call_packed put the mode back although the call raised
            continue
        ws = torch.full((n,), 0xA5, device=DEV, dtype=torch.uint8)
        H_.call("fs_conv2d_pack", H_.ptr(w), *shape, transposed, H_.ptr(ws), n, None)
        a = run(w, ws, True)
        b = run(w, ws, True)
        assert torch.equal(a, ref) and torch.equal(b, ref), (choice, transposed)
        stale = run(w2, ws, True)                                # the pack is what the kernel reads
        assert torch.equal(stale, ref)
        fresh = run(w2, ws, False)                               # the default mode re-packs
        assert relerr(fresh, 1.5 * ref) <= 1e-5
        assert int(lib.fs_conv2d_ws_mode(0)) == 0


def test_weight_packs_follow_every_arena_rewrite():
    """ops._pack_for / ops.repack_weights against the pack-per-call path, in deterministic mode (bit-identical or wrong): three training
    steps (the prefetch after each optimiser step, the frozen-optimiser window, an eval pass between steps), then load_state_dict and a
    torch-side in-place edit of one weight -- every way the weights change has to invalidate the packs."""
    from fovealseg import train
    H_ = fovealseg.hip
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    batch = train.synthetic_batch(4, 256, 256, seed=5, device=dev)

    def run(persist):
        ops.PACK_PERSIST = persist
        module, nets = train.build_module(cfg, device=dev)
        module.train()
        optimizers = train.create_optimizers(nets, cfg)
        ops.DropoutState.seed, ops.DropoutState.step = 11, 0
        outs = []
        for it in range(3):
            outs.append(float(train.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=it)[0]))
        module.eval()
        outs.append(float(train.eval_step(module, batch)[0]))
        module.train()
        outs.append(float(train.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=3)[0]))
        frozen = copy.deepcopy(cfg)                # deformation module frozen: only two of the four arenas are rewritten by this step
        frozen.TRAIN.fix_deform_aft_pretrain, frozen.TRAIN.fix_deform_start_epoch, frozen.TRAIN.fix_deform_end_epoch = True, 1, 1
        outs.append(float(train.train_step(module, optimizers, batch, frozen, epoch=1, cur_iter=4)[0]))
        outs.append(float(train.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=5)[0]))
        sd = {k: v.clone() for k, v in module.state_dict().items()}
        for k in sd:
            if sd[k].dtype == torch.float32 and sd[k].dim() == 4:
                sd[k] = sd[k] * 0.5
        module.load_state_dict(sd)
        for o in optimizers:
            o.flat.refresh_amax()          # what train.load_checkpoint does after rewriting the parameters
        module.eval()
        outs.append(float(train.eval_step(module, batch)[0]))
        with torch.no_grad():
            next(p for p in module.encoder.parameters() if p.dim() == 4 and p.shape[-1] == 3).mul_(2.0)      # torch-side edit: version counter
        outs.append(float(train.eval_step(module, batch)[0]))
        torch.cuda.synchronize()
        npacks = sum(len(p.__dict__.get("_fs_packs", ())) for p in module.parameters())
        return outs, [o.flat.data.clone() for o in optimizers], npacks

    assert not H_.get_deterministic()
    H_.set_deterministic(True)
    keep = ops.PACK_PERSIST
    try:
        la, pa, na = run(True)
        lb, pb, nb = run(False)
    finally:
        ops.PACK_PERSIST = keep
        H_.set_deterministic(False)
    assert na > 300 and nb == 0, (na, nb)          # the packs exist (forward and bwd-data of every conv layer) / the switch turns them off
    assert la == lb, (la, lb)
    assert len(set(la)) == len(la), la             # every stage really changed the result
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)


def test_static_weight_packs_for_serving():
    """ops.static_weight_packs: a serving process (no optimiser) keeps every conv layer's pack after the first forward; the results are
    those of the pack-per-call path, before and after the weights are replaced through load_state_dict."""
    from fovealseg import train
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    batch = train.synthetic_batch(2, 256, 256, seed=7, device=dev)
    module, _ = train.build_module(cfg, device=dev)
    module.eval()
    assert not ops.PACK_PERSIST
    ref0 = [float(v) for v in train.eval_step(module, batch)]
    sd = {k: (v * 0.5 if v.dtype == torch.float32 and v.dim() == 4 else v.clone()) for k, v in module.state_dict().items()}
    keep = {k: v.clone() for k, v in module.state_dict().items()}
    module.load_state_dict(sd)
    ref1 = [float(v) for v in train.eval_step(module, batch)]
    module.load_state_dict(keep)
    assert sum(len(p.__dict__.get("_fs_packs", ())) for p in module.parameters()) == 0
    ops.static_weight_packs(module)
    a = [float(v) for v in train.eval_step(module, batch)]
    n1 = sum(len(p.__dict__.get("_fs_packs", ())) for p in module.parameters())
    b = [float(v) for v in train.eval_step(module, batch)]           # second forward: every pack is reused
    assert n1 > 100 and n1 == sum(len(p.__dict__.get("_fs_packs", ())) for p in module.parameters())
    assert a == ref0 and b == ref0, (a, b, ref0)
    module.load_state_dict(sd)
    c = [float(v) for v in train.eval_step(module, batch)]
    assert c == ref1 and ref1 != ref0, (c, ref1, ref0)
    ops.static_weight_packs(module, on=False)
    assert sum(len(p.__dict__.get("_fs_packs", ())) for p in module.parameters()) == 0
    assert [float(v) for v in train.eval_step(module, batch)] == ref1


# Production spatial sizes: many pixel tiles per image, image borders inside tiles, the stacked-batch tiling of the
# small maps -- the multi-tile paths the <= 24x24 cases above barely touch.  fp64 reference, max-norm over every element.
CONV_CASES_FULLRES = [
    (8, 80, 80, 64, 64, 3, 1),       # halo-tiled 3x3, 8x16 patches
    (8, 40, 40, 128, 128, 3, 1),     # halo-tiled, 128-column workgroups
    (16, 20, 20, 256, 256, 3, 1),    # stacked-batch tiling
    (16, 10, 10, 512, 512, 3, 1),
    (8, 80, 80, 64, 128, 3, 2),      # tap-class kernel, stride 2 (four classes)
    (8, 80, 80, 64, 256, 1, 1),      # plain kernel, 1x1
    (4, 80, 80, 960, 240, 3, 1),     # C1 cbr: 30 K-chunks
    (64, 80, 80, 64, 64, 3, 2),      # round 4, at the bench batch: stride-2 forward over parity planes (64 columns), bwd-data with four parities per
                                     # workgroup, bwd-weight as nine gathered-row GEMMs
    (64, 20, 20, 256, 512, 3, 2),    # ... 128-column forward, bwd-weight on the nine-accumulator kernel + per-split slabs + ordered reduce
    (16, 80, 80, 960, 512, 3, 4),    # the C1 classification head's stride-4 conv: bwd-weight as nine gathered-row GEMMs
]


@pytest.mark.parametrize("case", CONV_CASES_FULLRES)
def test_conv_fwd_bwd_production_sizes(case, prec):
    B, H, W, Ci, Co, k, s = case
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    pad = k // 2
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, pad)
    cot = torch.randn(y64.shape, generator=g)
    y64.backward(cot.double())
    xd, wd, dyd = nhwc(x), rsck_param(w), nhwc(cot)
    y = ops.conv2d_fwd(xd, wd, None, s, pad)
    assert relerr(nchw(y), y64.detach()) <= 1e-5
    dx = ops.conv2d_bwd_data(dyd, wd, xd.shape, s, pad)
    assert relerr(nchw(dx), x64.grad) <= 1e-5
    dw = ops.conv2d_bwd_weight(xd, dyd, w.shape, s, pad)
    assert relerr(dw.cpu(), w64.grad) <= 2e-5


# Weight gradient of a linear layer / 1x1 convolution (csrc/conv_wgrad.hip linear_wgrad_kernel, bf16x3): rows x Cin x Cout chosen to reach
# every tile (64x64, 64x128, 128x64, 128x128 needs >= 512 rows per split), ragged channel counts (not multiples of 32), ragged row
# counts (not multiples of the 32-row chunk), fewer rows than one chunk, and accumulation into an existing gradient.
LINEAR_WGRAD_CASES = [(1000, 48, 96), (4100, 64, 64), (333, 64, 200), (2050, 200, 64), (70001, 160, 136), (17, 320, 1280), (6400, 512, 512),
                      (1600, 1280, 320), (25600, 320, 324)]


@pytest.mark.parametrize("rows,Ci,Co", LINEAR_WGRAD_CASES)
def test_linear_weight_gradient(rows, Ci, Co):
    fovealseg.hip.set_conv_precision("bf16x3")
    try:
        g = torch.Generator().manual_seed(rows + Ci + Co)
        x = torch.randn(rows, Ci, generator=g)
        dy = torch.randn(rows, Co, generator=g)
        ref = (x.double().t() @ dy.double())                                  # [Ci][Co]
        xd, dyd = x.to(DEV).view(1, rows, 1, Ci), dy.to(DEV).view(1, rows, 1, Co)
        dw = ops.conv2d_bwd_weight(xd, dyd, (Co, Ci, 1, 1), 1, 0)             # (Co,Ci,1,1) view of the RSCK buffer
        got = dw[:, :, 0, 0].t().double().cpu()
        # 24-bit operands (2^-24 per factor, random over `rows` products) and fp32 accumulation, the split-K partial sums added by
        # atomics at the magnitude of the running sum: measured 4e-7 of max |dW| on these shapes
        assert relerr(got, ref) <= 3e-6
        # accumulate = True adds to what the buffer holds
        buf = ops.new_rsck_weight(Co, Ci, 1, 1, device=DEV)
        ops.rsck(buf).fill_(1.5)
        ops.conv2d_bwd_weight(xd, dyd, (Co, Ci, 1, 1), 1, 0, out=buf, accumulate=True)
        got2 = buf[:, :, 0, 0].t().double().cpu()
        assert relerr(got2 - 1.5, ref) <= 2e-5            # (each atomic add now rounds at the magnitude of the running sum)
        # the fused entry point: the same dW plus the bias gradient (column sums of dy) from the one launch
        assert fovealseg.hip.linear_bwd_weight_bias_ok(rows, Ci, Co)
        dw3 = torch.empty(Ci, Co, device=DEV)
        db3 = torch.full((Co,), 0.25, device=DEV)
        fovealseg.hip.call("fs_linear_bwd_weight_bias", fovealseg.hip.ptr(xd), fovealseg.hip.ptr(dyd), fovealseg.hip.ptr(dw3),
                           fovealseg.hip.ptr(db3), rows, Ci, Co, 0, 1, None, 0)
        assert relerr(dw3.double().cpu(), ref) <= 3e-6
        assert relerr(db3.double().cpu() - 0.25, dy.double().sum(0)) <= 3e-6
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


SPLIT_SHAPES = [(2, 24, 24, 64, 64, 1), (1, 20, 20, 128, 96, 1), (2, 24, 24, 64, 128, 2)]   # halo 3x3, 2-chunk 3x3, tap-class stride 2


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("shape", SPLIT_SHAPES)
def test_split_precision_range_properties(mode, shape):
    """Size-independent properties of the scaled split-precision kernels: exact homogeneity under power-of-two
    scaling over 80 binades, zeros in -> zeros out, outliers, non-finite inputs."""
    fovealseg.hip.set_conv_precision(mode)
    try:
        B, H, W, Ci, Co, s = shape
        g = torch.Generator().manual_seed(7)
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
        Ho = (H + 2 - 3) // s + 1
        dy = torch.randn(B, Co, Ho, Ho, generator=g)
        xd, wd, dyd = nhwc(x), rsck_param(w), nhwc(dy)
        y0 = ops.conv2d_fwd(xd, wd, None, s, 1)
        dx0 = ops.conv2d_bwd_data(dyd, wd, xd.shape, s, 1)
        dw0 = ops.conv2d_bwd_weight(xd, dyd, w.shape, s, 1)
        ref = F.conv2d(x.double(), w.double(), None, s, 1)
        assert relerr(nchw(y0).double(), ref) <= 2e-6
        # 1. y(2^a x, 2^b w) = 2^(a+b) y(x, w) bit for bit: the tile / tensor exponents absorb the scale exactly
        for a, b in ((-40, 0), (30, -35), (0, 40), (-20, -20)):
            xa, wb = xd * 2.0 ** a, rsck_param(w * 2.0 ** b)
            assert torch.equal(ops.conv2d_fwd(xa, wb, None, s, 1), y0 * 2.0 ** (a + b)), (a, b)
            assert torch.equal(ops.conv2d_bwd_data(dyd * 2.0 ** a, wb, xd.shape, s, 1), dx0 * 2.0 ** (a + b)), (a, b)
            # bwd-weight sums its split-K partials with float atomics, whose order varies from launch to launch: same value up to that
            assert relerr(ops.conv2d_bwd_weight(xa, dyd * 2.0 ** b, w.shape, s, 1), dw0 * 2.0 ** (a + b)) <= 2e-6, (a, b)
        # 2. zeros in -> exact zeros out (all-zero tiles have no exponent of their own)
        z = torch.zeros_like(xd)
        assert float(ops.conv2d_fwd(z, wd, None, s, 1).abs().max()) == 0.0
        assert float(ops.conv2d_bwd_weight(z, dyd, w.shape, s, 1).abs().max()) == 0.0
        half = xd.clone(); half[0] = 0                          # one image all zero, the other not
        yh = ops.conv2d_fwd(half, wd, None, s, 1)
        assert float(yh[0].abs().max()) == 0.0 and torch.equal(yh[1:], y0[1:])
        # 3. one outlier 2^30 above the rest: pixels outside its 3x3 reach keep an error far below their own magnitude
        xo = x.clone(); xo[0, 3, 5, 5] = 2.0 ** 30
        yo = nchw(ops.conv2d_fwd(nhwc(xo), wd, None, s, 1)).double()
        ro = F.conv2d(xo.double(), w.double(), None, s, 1)
        lib = fovealseg.hip.load()
        wsb = fovealseg.hip.conv_workspace_bytes(H, W, Ci, Ho, Ho, Co, 3, 3, s, 1, 1, 0)
        choice = lib.fs_conv2d_kernel_choice(B, H, W, Ci, Ho, Ho, Co, 3, 3, s, 1, 1, 0, wsb)
        mask = torch.ones_like(ro, dtype=torch.bool)
        lo, hi = (5 - 1) // s, (5 + 1) // s
        mask[0, :, max(lo, 0):hi + 1, max(lo, 0):hi + 1] = False
        if choice == 8:
            # F(4,3): the outlier enters the transform of its whole quad (output columns 4..7 for input column 5); in exact arithmetic it
            # cancels out of column 7, in fp32 it leaves a transform-domain error there -- that column belongs to the reach
            mask[0, :, max(lo, 0):hi + 1, 4:8] = False
        assert float((yo - ro)[mask].abs().max()) <= 2.0 ** 30 * 2.0 ** -36     # <= 2^-36 of the tile maximum (bound 2^-39 per term)
        if choice == 5:
            # inside the reach the F(2,3) kernel forms the output from transform-domain products of size outlier * |U| (U = sums
            # of the filter row), so its error is relative to outlier * max|w|, not to the individual (possibly tiny) output
            assert float((yo - ro)[~mask].abs().max()) <= 2.0 ** 30 * float(w.abs().max()) * 2e-7
        elif choice == 8:
            # the F(4,3) kernel likewise, with transform coefficients up to 5 in, 8 out
            assert float((yo - ro)[~mask].abs().max()) <= 2.0 ** 30 * float(w.abs().max()) * 4e-6
        else:
            # every direct kernel (halo, tap-class, plain) keeps the per-output bound (VERDICT r2 #5c: keyed on the kernel that ran)
            assert choice in (2, 3, 4, 1, 7), choice            # 7 = the stride-2 forward over parity planes: direct products as well
            assert float(((yo - ro)[~mask].abs() / ro[~mask].abs().clamp_min(1)).max()) <= 1e-5
        # 4. non-finite inputs propagate (no hang, no silent number)
        xn = xd.clone(); xn[0, 2, 2, 0] = float("inf")
        yn = ops.conv2d_fwd(xn, wd, None, s, 1)
        assert not torch.isfinite(yn[0]).all() and torch.isfinite(yn[1:]).all()
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


def test_maxpool_and_dropout():
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 16, 13, 11, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    xd = nhwc(x).requires_grad_(True)
    out = ops.MaxPool.apply(xd, 3, 2, 1)
    out.backward(nhwc(cot))
    assert torch.equal(nchw(out), ref.detach())
    assert relerr(nchw(xd.grad), xr.grad) <= 1e-6
    key = ops.layer_key(9, 99)
    v = torch.randn(3, 5, 7, 8, generator=g)
    vd = v.to(DEV).requires_grad_(True)
    o = ops.Dropout.apply(vd, 0.5, key)
    o.backward(torch.ones_like(o))
    keep = torch.from_numpy(O.dropout_keep_mask_nhwc(v.numel(), key, 0.5)).view(v.shape)
    assert torch.equal(o.detach().cpu(), torch.where(keep, v * 2.0, torch.zeros(())))
    assert torch.equal(vd.grad.cpu(), keep.float() * 2.0)


def test_conv_rejects_bad_shapes():
    x = torch.zeros(1, 4, 4, 4, device=DEV)
    w = ops.new_rsck_weight(4, 4, 3, 3, device=DEV)
    y = torch.zeros(1, 5, 5, 4, device=DEV)       # wrong output size
    with pytest.raises(fovealseg.hip.HipLibraryError):
        fovealseg.hip.call("fs_conv2d_fwd", x.data_ptr(), ops.rsck(w).data_ptr(), None, y.data_ptr(), 1, 4, 4, 4, 5, 5, 4, 3, 3, 1, 1, 1, 0.0, 0, None, 0, None)
    with pytest.raises(fovealseg.hip.HipLibraryError):
        ops.conv2d_fwd(torch.zeros(1, 4, 4, 4), w, None, 1, 1)   # CPU tensor: no fallback


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act,use_res,drop", [(1, True, 0.3), (2, False, 0.0), (0, False, 0.0)])
def test_conv_bn_act(training, act, use_res, drop, prec):
    B, C, H, W = 3, 64, 12, 12
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / 24.0
    gamma = 0.5 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    rm = 0.1 * torch.randn(C, generator=g)
    rv = 1 + 0.1 * torch.rand(C, generator=g)
    res = torch.randn(B, C, H, W, generator=g) if use_res else None
    key = ops.layer_key(123, 456)

    # ---- HIP side ----
    xd = nhwc(x).requires_grad_(True)
    wd = rsck_param(w).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rd = nhwc(res).requires_grad_(True) if use_res else None
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    cot = torch.randn(B, C, H, W, generator=g)

    class Cnt:
        n = 0

        def add_(self, k):
            self.n += k
    meta = dict(stride=1, pad=1, act=act, training=training, momentum=0.1, drop_p=drop, drop_key=key,
                running_mean=rmd, running_var=rvd, num_batches_tracked=Cnt())
    zd = ops.ConvBnAct.apply(xd, wd, None, gd, bd, rd, meta)
    zd.backward(nhwc(cot))
    zh = nchw(zd)

    # ---- oracle side (torch CPU); the activation takes the branch the device took (see `replay`) ----
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if use_res else None
    y = F.conv2d(xr, wr, None, 1, 1)
    if training and drop > 0:
        keep = O.dropout_keep_mask_nhwc(B * H * W * C, key, drop).reshape(B, H, W, C)
        mask = torch.from_numpy(keep).permute(0, 3, 1, 2).float() * np.float32(1.0 / (1.0 - drop))
        y = y * mask
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(y, rm_ref, rv_ref, gr, br, training, 0.1, 1e-5)
    if use_res:
        z = z + rr
    if act == 1:
        z = z * (zh > 0).float()
    elif act == 2:
        z = z * ((zh > 0) & (zh < 6)).float() + 6.0 * (zh >= 6).float()
    z.backward(cot)

    assert relerr(zh, z.detach()) <= 2e-5
    assert relerr(nchw(xd.grad), xr.grad) <= 1e-4
    assert relerr(wd.grad.cpu(), wr.grad) <= 1e-4
    assert relerr(gd.grad.cpu(), gr.grad) <= 1e-4
    assert relerr(bd.grad.cpu(), br.grad) <= 1e-4
    if use_res:
        assert relerr(nchw(rd.grad), rr.grad) <= 1e-5
    if training:
        assert relerr(rmd.cpu(), rm_ref) <= 1e-5 and relerr(rvd.cpu(), rv_ref) <= 1e-5
        assert meta["num_batches_tracked"].n == 1


# ------------------------------------------------------------------------------------------------
# front-end
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,Ci,Co", [(3, 64, 64), (1, 64, 128)])
def test_conv_batch_ranges_for_tensors_beyond_4gb(k, Ci, Co, monkeypatch):
    """Tensors of 4 GB or more are convolved in batch ranges (ops._batch_ranges: the kernels address with 32-bit byte offsets).  With the
    limit lowered so that B = 7 splits into ranges of 2, conv + BatchNorm forward and all gradients must equal the unsplit call: outputs
    bit for bit (an output element's sum does not depend on its image's position in the batch), statistics / weight gradient to the
    reordering of their sums."""
    fovealseg.hip.set_conv_precision("bf16x3")
    try:
        B, H, W = 7, 12, 10
        g = torch.Generator().manual_seed(k + Ci)
        x = torch.randn(B, H, W, Ci, generator=g).to(DEV)
        w = rsck_param(torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5)
        gamma, beta = (1 + 0.1 * torch.randn(Co, generator=g)).to(DEV), (0.1 * torch.randn(Co, generator=g)).to(DEV)
        cot = torch.randn(B, H, W, Co, generator=g).to(DEV)

        class Cnt:
            def add_(self, n):
                pass

        def run():
            xd, wd, gd, bd = (t.clone().requires_grad_(True) for t in (x, w, gamma, beta))
            meta = dict(stride=1, pad=k // 2, act=1, training=True, momentum=0.1, drop_p=0.0, drop_key=0,
                        running_mean=torch.zeros(Co, device=DEV), running_var=torch.ones(Co, device=DEV), num_batches_tracked=Cnt())
            z = ops.ConvBnAct.apply(xd, wd, None, gd, bd, None, meta)
            z.backward(cot)
            return z.detach(), xd.grad, wd.grad, gd.grad, bd.grad, meta["running_mean"], meta["running_var"]
        whole = run()
        per_image = max(H * W * Ci, H * W * Co) * 4
        monkeypatch.setattr(ops, "MAX_TENSOR_BYTES", 2 * per_image + 1)
        assert ops._batch_ranges(B, H * W * Ci, H * W * Co) == [(0, 2), (2, 4), (4, 6), (6, 7)]
        split = run()
        assert relerr(split[0], whole[0]) <= 1e-6          # z: the batch statistics differ in the last bits (slab order)
        assert relerr(split[1], whole[1]) <= 1e-5
        for a, b_ in zip(split[2:], whole[2:]):
            assert relerr(a, b_) <= 1e-5
        # the three convolutions themselves: bit-identical outputs, range by range
        y_w = ops.conv2d_fwd(x, w, None, 1, k // 2)
        dx_w = ops.conv2d_bwd_data(cot, w, x.shape, 1, k // 2)
        monkeypatch.setattr(ops, "MAX_TENSOR_BYTES", 4294967000)
        assert torch.equal(y_w, ops.conv2d_fwd(x, w, None, 1, k // 2))
        assert torch.equal(dx_w, ops.conv2d_bwd_data(cot, w, x.shape, 1, k // 2))
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


def test_atrous_conv_as_phase_batches_and_centre_tap(prec):
    """modules.conv_bn_act's two rewrites of an atrous 3x3 conv, against F.conv2d with the dilation in fp64: (a) dilation d, pad d on a map
    divisible by d = ordinary 3x3 convs on the d*d phase images (space-to-batch), forward and both gradients; (b) a rate at least as large
    as the map = the 1x1 conv of the centre tap, with zero gradient on the eight other taps."""
    from fovealseg import modules as M
    g = torch.Generator().manual_seed(77)
    B, C, Co, H, W, d = 2, 32, 48, 12, 8, 2
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, None, 1, d, d)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot.double())
    xd, wd = nhwc(x).requires_grad_(True), rsck_param(w).requires_grad_(True)
    y = M._batch_to_space(ops.ConvBias.apply(M._space_to_batch(xd, d), wd, None, 1, 1), d)
    y.backward(nhwc(cot))
    assert relerr(nchw(y), ref.detach()) <= 2e-5
    assert relerr(nchw(xd.grad), x64.grad) <= 2e-5
    assert relerr(wd.grad.cpu(), w64.grad) <= 5e-5
    # (b) rate 12 on a 10 x 10 map
    x = torch.randn(B, C, 10, 10, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, None, 1, 12, 12)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot.double())
    xd, wd = nhwc(x).requires_grad_(True), rsck_param(w).requires_grad_(True)
    y = ops.ConvBias.apply(xd, ops.CenterTap.apply(wd), None, 1, 0)
    y.backward(nhwc(cot))
    assert relerr(nchw(y), ref.detach()) <= 2e-5
    assert relerr(nchw(xd.grad), x64.grad) <= 2e-5
    assert relerr(wd.grad.cpu(), w64.grad) <= 5e-5
    assert float(wd.grad[:, :, 0, 0].abs().max()) == 0.0 and float(w64.grad[:, :, 0, 0].abs().max()) == 0.0


def test_gaze_lowres_g2(golden):
    g = golden("g2_lowres_128")
    out = ops.gaze_lowres(T(g["x"]).to(DEV), T(g["focus"]).to(DEV), 80, 80)
    assert np.abs(nchw(out).numpy() - g["x_low"]).max() <= 1e-6
    g = golden("g2_lowres_640")
    gen = torch.Generator().manual_seed(int(g["seed"]))
    X = torch.rand(2, 3, 640, 640, generator=gen)
    out = ops.gaze_lowres(X.to(DEV), T(g["focus"]).to(DEV), 80, 80)
    assert np.abs(nchw(out).numpy() - g["x_low"]).max() <= 1e-6


def test_compress_softmax():
    g = torch.Generator().manual_seed(3)
    s = torch.randn(3, 24, 80, 80, generator=g)
    w = torch.randn(1, 24, 1, 1, generator=g) * 0.3
    b = torch.randn(1, generator=g)
    sr, wr, br = s.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xs_ref = F.softmax(F.conv2d(F.relu(sr), wr, br).view(3, -1), 1).view(3, 1, 80, 80)
    cot = torch.randn(xs_ref.shape, generator=g)
    xs_ref.backward(cot)
    sd = nhwc(s).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    xs = ops.CompressSoftmax.apply(sd, wd, bd)
    xs.backward(cot.to(DEV))
    assert np.abs(xs.detach().cpu().numpy() - xs_ref.detach().numpy()).max() <= 1e-7
    assert abs(float(xs.sum()) - 3.0) <= 1e-4
    assert relerr(nchw(sd.grad), sr.grad) <= 1e-4
    assert relerr(wd.grad.cpu(), wr.grad) <= 1e-4
    assert abs(float(bd.grad) - float(br.grad)) <= 1e-6      # softmax is shift-invariant: d/db == 0


def test_compress_net_forward_returns_logits_like_the_reference():
    """CompressNet.forward as the reference calls it (models/models.py:713, 360-372): (B,24,H,W) -> (B,1,H,W) logits, forward and
    every gradient against oracle.OracleCompress with the same weights; and softmax over it == the fused softmax_nhwc path."""
    g = torch.Generator().manual_seed(11)
    cfg = fovealseg.lvis50_cfg()
    comp = fovealseg.ModelBuilder.build_net_compress(cfg)
    fovealseg.weights.apply_name_keyed_init(comp, "net_compress.")
    ref = O.OracleCompress(24)
    ref.load_state_dict({k: v.detach().clone() for k, v in comp.state_dict().items()})
    comp.to(DEV)
    for shape in ((3, 24, 80, 80), (2, 24, 37, 53)):
        x = torch.randn(*shape, generator=g)
        cot = torch.randn(shape[0], 1, *shape[2:], generator=g)
        xr = x.clone().requires_grad_(True)
        ref.zero_grad()
        out_ref = ref(xr)
        out_ref.backward(cot)
        xd = x.to(DEV).requires_grad_(True)
        comp.zero_grad()
        out = comp(xd)
        assert out.shape == out_ref.shape
        out.backward(cot.to(DEV))
        assert np.abs(out.detach().cpu().numpy() - out_ref.detach().numpy()).max() <= 2e-6
        assert relerr(xd.grad.cpu(), xr.grad) <= 1e-6
        assert relerr(comp.conv_last.weight.grad.cpu(), ref.conv_last.weight.grad) <= 2e-5
        assert relerr(comp.conv_last.bias.grad.cpu(), ref.conv_last.bias.grad) <= 2e-5
        with torch.no_grad():
            fused = comp.softmax_nhwc(nhwc(x))
            plain = F.softmax(comp(x.to(DEV)).view(shape[0], -1), 1).view_as(fused)
        assert float((fused - plain).abs().max()) <= 1e-7


def test_g3_saliency_stack(golden):
    g = golden("g3_saliency")
    cfg = fovealseg.lvis50_cfg()
    sal = fovealseg.ModelBuilder.build_net_saliency(cfg)
    comp = fovealseg.ModelBuilder.build_net_compress(cfg)
    for mode in ("eval", "train"):
        fovealseg.weights.apply_name_keyed_init(sal, "localization.")
        fovealseg.weights.apply_name_keyed_init(comp, "net_compress.")
        sal.to(DEV).train(mode == "train")
        with torch.no_grad():
            xs = comp.to(DEV).softmax_nhwc(sal.forward_nhwc(nhwc(T(g["x_low"]))))
        assert np.abs(xs.cpu().numpy() - g["xs_" + mode]).max() <= 2e-7


def test_area_pool_and_edge_loss_g10(golden):
    g = golden("g10_losses")
    gen = torch.Generator().manual_seed(int(g["y_seed"]))
    torch.rand(3, 3, 256, 256, generator=gen)
    Fp = torch.rand(3, 2, generator=gen) * 0.8 + 0.1
    ii = torch.arange(256, dtype=torch.float32)
    cy, cx = (Fp[:, 0] * 255)[:, None, None], (Fp[:, 1] * 255)[:, None, None]
    Y = (((ii[None, :, None] - cy) ** 2 + (ii[None, None, :] - cx) ** 2) <= (0.15 * 256) ** 2).float().unsqueeze(1)
    t = ops.area_pool(Y.to(DEV), 80, 80)
    assert np.abs(t.cpu().numpy() - g["area"]).max() <= 1e-6
    # non-binary, non-divisible sizes
    R = torch.rand(2, 1, 333, 517)
    tr = ops.area_pool(R.to(DEV), 80, 80)
    assert np.abs(tr.cpu().numpy() - F.interpolate(R, size=(80, 80), mode="area").numpy()).max() <= 1e-6
    xs = T(g["xs"]).to(DEV).requires_grad_(True)
    el = ops.EdgeLoss.apply(xs, t, 0.05 * 100.0)
    el.backward()
    assert abs(float(el) - float(g["edge"])) <= 1e-6
    assert relerr(xs.grad.cpu(), T(g["dxs"])) <= 1e-4


def test_gauss_grid_g4(golden):
    g = golden("g4_grid")
    g1d = torch.from_numpy(O.gaussian_1d(91, 45)).to(DEV)
    xs = T(g["xs"]).to(DEV).requires_grad_(True)
    grid = ops.GaussGrid.apply(xs, g1d, 45)
    got = grid.detach().cpu().numpy()
    assert np.abs(got - g["grid"]).max() <= 3e-5          # reference fp32 (itself 1.75e-5 from fp64)
    x64 = T(g["xs"]).double().requires_grad_(True)
    g64 = O.create_grid_f64(x64, 45)
    assert np.abs(got - g64.detach().numpy()).max() <= 3e-6    # fp64 evaluation of the same formula
    assert got.min() >= -1.0 and got.max() <= 1.0
    # backward, random-saliency samples: well conditioned, must match the reference's own autograd
    grid.backward(T(g["cot"]).to(DEV))
    assert relerr(xs.grad.cpu()[:2], T(g["dxs"])[:2]) <= 1e-4
    # backward, all samples, against the fp64 evaluation.  A uniform background + replication padding
    # puts every border grid point EXACTLY on the clamp bound (centroid of a symmetric window), where
    # rounding noise decides the clamp mask (the reference's own fp32 gradient is 12 % from fp64 there);
    # those points get a zero cotangent so the comparison is well conditioned.
    safe = ((g64.detach().abs() - 1).abs() > 1e-4).to(torch.float64)
    cot = T(g["cot"]).double() * safe
    (g64 * cot).sum().backward()
    xs2 = T(g["xs"]).to(DEV).requires_grad_(True)
    ops.GaussGrid.apply(xs2, g1d, 45).backward(cot.float().to(DEV))
    assert relerr(xs2.grad.cpu(), x64.grad.float()) <= 1e-4


@pytest.mark.parametrize("tag", ["128x128", "200x136"])
def test_grid_sample_g5_bitexact(golden, tag):
    g = golden("g5_gridsample_" + tag)
    x, y, grid = T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["grid"]).to(DEV).requires_grad_(True)
    out = ops.GridSample.apply(x, grid)
    assert np.array_equal(nchw(out).numpy(), g["x_sampled"])             # bit-exact fp32
    label, ys = ops.grid_sample_label(y, grid.detach(), return_float=True)
    assert np.array_equal(ys.cpu().numpy(), g["y_sampled"])
    assert np.array_equal(label.cpu().numpy(), g["label"])                 # bit-exact int64
    out.backward(nhwc(T(g["cot"])))
    assert np.abs(grid.grad.cpu().numpy() - g["dgrid"]).max() <= 1e-6 * max(1.0, np.abs(g["dgrid"]).max())


def test_grid_sample_bwd_input():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 40, 56, generator=g)
    grid = torch.rand(2, 16, 16, 2, generator=g) * 2.2 - 1.1
    xr = x.clone().requires_grad_(True)
    out_ref = F.grid_sample(xr, grid, align_corners=False)
    cot = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(cot)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.GridSample.apply(xd, grid.to(DEV))
    out.backward(nhwc(cot))
    assert relerr(xd.grad.cpu(), xr.grad) <= 1e-5


@pytest.mark.parametrize("H", [128, 640])
def test_inverse_index_maps_g6(golden, H):
    g = golden(f"g6_inverse_{H}")
    u, v = ops.inverse_index_maps(T(g["grid"]).to(DEV), H, H)
    assert np.array_equal(u.cpu().numpy(), g["u"]) and np.array_equal(v.cpu().numpy(), g["v"])


# ------------------------------------------------------------------------------------------------
# fuse / concat / head / loss
# ------------------------------------------------------------------------------------------------
def test_hr_fuse_and_concat():
    g = torch.Generator().manual_seed(11)
    B = 2
    xs = [torch.randn(B, 32, 16, 16, generator=g), torch.randn(B, 32, 8, 8, generator=g),
          torch.randn(B, 32, 4, 4, generator=g), torch.randn(B, 32, 2, 2, generator=g)]
    xr = [t.clone().requires_grad_(True) for t in xs]
    y = xr[0]
    for t in xr[1:]:
        y = y + F.interpolate(t, size=(16, 16), mode="bilinear", align_corners=False)
    y = F.relu(y)
    cot = torch.randn(y.shape, generator=g)
    y.backward(cot)
    xd = [nhwc(t).requires_grad_(True) for t in xs]
    out = ops.HrFuse.apply(16, 16, *xd)
    out.backward(nhwc(cot))
    assert relerr(nchw(out), y.detach()) <= 1e-6
    for a, b in zip(xd, xr):
        assert relerr(nchw(a.grad), b.grad) <= 1e-5
    # concat
    xr = [t.clone().requires_grad_(True) for t in xs]
    cat = torch.cat([xr[0]] + [F.interpolate(t, size=(16, 16), mode="bilinear", align_corners=False) for t in xr[1:]], 1)
    cot = torch.randn(cat.shape, generator=g)
    cat.backward(cot)
    xd = [nhwc(t).requires_grad_(True) for t in xs]
    out = ops.UpsampleConcat.apply(*xd)
    out.backward(nhwc(cot))
    assert relerr(nchw(out), cat.detach()) <= 1e-6
    for a, b in zip(xd, xr):
        assert relerr(nchw(a.grad), b.grad) <= 1e-5


def test_seg_loss_g10(golden):
    g = golden("g10_losses")
    pred = T(g["pred"]).to(DEV).requires_grad_(True)
    gt = T(g["gt"]).to(DEV)
    out = ops.SegLoss.apply(pred, gt, 5.0)
    out[0].backward()
    o = out.detach().cpu().numpy()
    assert abs(o[1] - float(g["focal"])) <= 1e-6 and abs(o[2] - float(g["dice"])) <= 1e-6
    assert np.abs(o[3:7] - g["accs"]).max() <= 1e-6
    assert np.abs(pred.grad.cpu().numpy() - g["dpred"]).max() <= 1e-6 * max(1.0, np.abs(g["dpred"]).max())


@pytest.mark.parametrize("B,K,H,W", [(2, 51, 37, 41), (3, 7, 5, 9)])
def test_seg_loss_ragged_sizes(B, K, H, W):
    # pixel counts that are no multiple of the 64-lane waves / 256-thread blocks (partly idle waves in the class sums)
    gen = torch.Generator().manual_seed(K * H)
    pred = torch.randn(B, K, H, W, generator=gen) * 2
    gt = torch.randint(0, K, (B, H, W), generator=gen)
    gt[0, : H // 2] = K - 1
    pd = pred.to(DEV).requires_grad_(True)
    out = ops.SegLoss.apply(pd, gt.to(DEV), 5.0)
    out[0].backward()
    pr = pred.clone().requires_grad_(True)
    fl, dl = O.focal_loss(pr, gt, 5.0), O.dice_loss_multiclass(pr, gt)
    (fl + dl).backward()
    o = out.detach().cpu().numpy()
    assert abs(o[1] - float(fl)) <= 2e-6 and abs(o[2] - float(dl)) <= 2e-6
    accs = [float(a) for a in O.accuracies(pred, gt, bg=K - 1)]
    assert np.abs(o[3:7] - np.array(accs)).max() <= 1e-6
    assert relerr(pd.grad.cpu(), pr.grad) <= 1e-5


@pytest.mark.parametrize("shape", [(1, 1, 9, 11), (2, 1, 13, 7), (3, 1, 80, 80), (64, 1, 80, 80), (33, 1, 79, 81)])
def test_edge_loss_ragged_sizes(shape):
    # element counts with a scalar tail behind the 16-byte passes; round 5: one, five, 100 and 52 workgroups per pass (whole-batch min /
    # max and sums as per-workgroup partials + an ordered sum), the last with a tail
    gen = torch.Generator().manual_seed(shape[2])
    xs = torch.rand(shape, generator=gen)
    t = torch.rand(shape, generator=gen)
    xd = xs.to(DEV).requires_grad_(True)
    el = ops.EdgeLoss.apply(xd, t.to(DEV), 5.0)
    el.backward()
    xr = xs.clone().requires_grad_(True)
    a = (xr - xr.min()) / (xr.max() - xr.min())
    b = (t - t.min()) / (t.max() - t.min())
    ref = 5.0 * F.mse_loss(a, b)
    ref.backward()
    assert abs(float(el) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
    assert relerr(xd.grad.cpu(), xr.grad) <= 1e-5


@pytest.mark.parametrize("hs,ws,pad", [(48, 100, 45), (32, 160, 20), (80, 80, 45), (50, 37, 30), (40, 3, 45)])
def test_gauss_grid_bwd_shapes(hs, ws, pad):
    # non-square grids: border-weight tables (sides <= 128) and the tap-by-tap border path (wider), against fp64 autograd; round 5: four
    # column bands per image -- widths that do not divide by four (37: bands of 10, 10, 10, 7) and fewer columns than bands (3)
    gen = torch.Generator().manual_seed(hs + ws)
    xs = torch.softmax(torch.randn(2, hs * ws, generator=gen) * 2, dim=1).view(2, 1, hs, ws)
    g1d = torch.from_numpy(O.gaussian_1d(2 * pad + 1, pad)).to(DEV)
    x64 = xs.double().requires_grad_(True)
    g64 = O.create_grid_f64(x64, pad)
    safe = ((g64.detach().abs() - 1).abs() > 1e-4).to(torch.float64)
    cot = torch.randn(2, hs, ws, 2, generator=gen).double() * safe
    (g64 * cot).sum().backward()
    xd = xs.to(DEV).requires_grad_(True)
    grid = ops.GaussGrid.apply(xd, g1d, pad)
    assert np.abs(grid.detach().cpu().numpy() - g64.detach().numpy()).max() <= 3e-6
    grid.backward(cot.float().to(DEV))
    assert relerr(xd.grad.cpu(), x64.grad.float()) <= 1e-4


def test_weight_amax_segments_unaligned():
    # parameters at arbitrary arena offsets: 16-byte body where the segment is aligned, scalar path and tails elsewhere
    gen = torch.Generator().manual_seed(11)
    sizes = [1, 3, 64, 255, 1000, 1025, 4099, 70001, 300000]
    offs, n = [], 5
    for i, sz in enumerate(sizes):
        offs.append(n)
        n += sz + (i % 3)
    arena = torch.randn(n, generator=gen)
    arena[offs[5] + 1024] = 77.0          # maximum in a tail element
    arena[offs[8] + 299999] = -91.0
    ad = arena.to(DEV)
    out = torch.empty(len(sizes), device=DEV, dtype=torch.int32)
    hip = fovealseg.hip
    offs_d = torch.tensor(offs, dtype=torch.int64, device=DEV)
    sizes_d = torch.tensor(sizes, dtype=torch.int64, device=DEV)
    hip.call("fs_weight_amax_segments", hip.ptr(ad), hip.ptr(offs_d), hip.ptr(sizes_d), len(sizes), hip.ptr(out))
    got = out.cpu().view(torch.float32)
    want = torch.stack([arena[o:o + s].abs().max() for o, s in zip(offs, sizes)])
    assert torch.equal(got, want)


@pytest.mark.parametrize("Ci,Co,k,stride", [(3, 64, 3, 1), (5, 192, 3, 1), (5, 24, 3, 2), (7, 32, 1, 1), (18, 20, 3, 1)])
def test_odd_input_channels_padded(Ci, Co, k, stride, prec):
    # layers whose input-channel count is no multiple of 4 run as zero-padded aligned problems (ops.pad_in_channels)
    gen = torch.Generator().manual_seed(Ci * 100 + Co)
    B, H, W, pad = 3, 19, 23, k // 2
    x = torch.randn(B, H, W, Ci, generator=gen)
    wl = torch.randn(Co, Ci, k, k, generator=gen) * 0.2
    w = ops.new_rsck_weight(Co, Ci, k, k, device=DEV)
    w.copy_(wl.to(DEV))
    w.requires_grad_(True)
    xd = x.to(DEV).requires_grad_(True)
    xp, wp = ops.pad_in_channels(xd, w)
    assert (xp.shape[-1] % 4 == 0 and xp.shape[-1] >= 16) == (prec != "f32") or Ci % 4 == 0
    y = ops.ConvBias.apply(xp, wp, None, stride, pad)
    x64 = x.permute(0, 3, 1, 2).double().requires_grad_(True)
    w64 = wl.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, stride, pad)
    cot = torch.randn(y64.shape, generator=gen)
    (y64 * cot.double()).sum().backward()
    (y * cot.permute(0, 2, 3, 1).to(DEV)).sum().backward()
    assert relerr(nchw(y.detach()), y64.detach().float()) <= 1e-5
    assert relerr(nchw(xd.grad), x64.grad.float()) <= 1e-5
    assert relerr(w.grad.cpu(), w64.grad.float()) <= 1e-5


def _hip_module():
    cfg = fovealseg.lvis50_cfg()
    MB = fovealseg.ModelBuilder
    m = fovealseg.DeformSegmentationModule(MB.build_encoder("hrnetv2_nodownsp", 960, ""), MB.build_decoder("c1", 960, 51, ""),
                                           MB.build_net_saliency(cfg), MB.build_net_compress(cfg), None, cfg)
    fovealseg.weights.apply_name_keyed_init(m)
    return m.to(DEV)


_PRISTINE = {}


def _restore(m):
    """Back to the name-keyed initial state.  The parity tests never step an optimiser on these shared modules, so the only
    tensors that move are the buffers (BatchNorm running statistics / counters in train-mode forwards): the first call keeps a
    copy of every buffer and a checksum of the parameters, later calls copy the buffers back (a full name-keyed re-init of the
    130 M parameters costs ~5 s per call, and the suite asks for it ~100 times)."""
    snap = _PRISTINE.get(id(m))
    if snap is None:
        fovealseg.weights.apply_name_keyed_init(m)
        bufs = {k: v.detach().clone() for k, v in m.named_buffers()}
        csum = float(sum(p.detach().double().sum() for p in m.parameters()))
        _PRISTINE[id(m)] = (m, bufs, csum)
        return
    _, bufs, csum = snap
    with torch.no_grad():
        for k, v in m.named_buffers():
            v.copy_(bufs[k])
    for mod in m.modules():
        if hasattr(mod, "_pending_batches"):
            mod._pending_batches = 0
    assert float(sum(p.detach().double().sum() for p in m.parameters())) == csum, "a test modified the shared module's parameters"


@pytest.fixture(scope="module")
def hipmod():
    m = _hip_module()
    _restore(m)
    return m


def _sub(m, path):
    for p in path.split("."):
        m = m[int(p)] if p.isdigit() else getattr(m, p)
    return m


def _set_drop(m, p):
    for d in m.modules():
        if hasattr(d, "drop_p"):
            d.drop_p = p


@pytest.fixture(scope="module")
def oracle():
    o = O.OracleDeformSeg()
    _restore(o)
    O.assign_paths(o)
    return o


def _run_oracle_block(oblk, name, rel, ins, train):
    ctx = O._Ctx(train, (lambda n, t: t) if train else None)       # Dropout(0.3) at p=0 in the goldens
    if name == "basic":
        return [oblk(ins[0], ctx, rel)]
    if name == "bottleneck":
        return [oblk(ins[0])]
    return oblk(ins, ctx, rel)


# One tolerance table for all three precision modes: with the activation branches replayed the backward is a
# smooth function on both sides, so f16x2 / bf16x3 meet the same bounds as the fp32-MFMA mode.
TOL_BLOCK_OUT, TOL_BLOCK_DIN, TOL_BLOCK_DW = 2e-5, 2e-4, 5e-4
TOL_E2E_GRAD = 1e-3          # rms-relative, per parameter tensor, through the full depth of the network (fwd ~100 layers + bwd)


@pytest.mark.parametrize("name", ["basic", "bottleneck", "hrmodule4"])
@pytest.mark.parametrize("mode", ["eval", "train_p0"])
def test_g7_blocks(golden, hipmod, oracle, name, mode, prec):
    g = golden(f"g7_{name}_{mode}")
    prefix = str(g["prefix"])
    _restore(hipmod)
    _restore(oracle)
    blk, oblk = _sub(hipmod, prefix), _sub(oracle, prefix)
    blk.train(mode != "eval")
    oblk.train(mode != "eval")
    _set_drop(blk, 0.0)
    n_in = sum(1 for k in g.files if k.startswith("in"))
    ins = [nhwc(T(g[f"in{i}"])).requires_grad_(True) for i in range(n_in)]
    with traced() as tr:
        outs = blk(ins[0]) if n_in == 1 else blk(ins)
    outs = [outs] if isinstance(outs, torch.Tensor) else list(outs)
    blk.zero_grad()
    torch.autograd.backward(outs, [nhwc(T(g[f"cot{i}"])) for i in range(len(outs))])
    for i, o in enumerate(outs):                          # forward: against the reference's own output
        ref = g[f"out{i}"]
        assert np.abs(nchw(o).numpy() - ref).max() <= TOL_BLOCK_OUT * max(1.0, np.abs(ref).max())
    # backward: against the oracle with the device's activation branches replayed (oracle pinned to these goldens on CPU)
    oins = [T(g[f"in{i}"]).clone().requires_grad_(True) for i in range(n_in)]
    with replay(hipmod, tr):
        oouts = _run_oracle_block(oblk, name, prefix.split(".", 1)[1], oins, mode != "eval")
        oblk.zero_grad()
        sum((o * T(g[f"cot{i}"])).sum() for i, o in enumerate(oouts)).backward()
    for i, o in enumerate(oouts):                         # the replay leaves the forward where the reference has it
        ref = g[f"out{i}"]
        assert np.abs(o.detach().numpy() - ref).max() <= TOL_BLOCK_OUT * max(1.0, np.abs(ref).max())
    for i, t in enumerate(ins):
        assert relerr(nchw(t.grad), oins[i].grad) <= TOL_BLOCK_DIN, i
        # and the replayed oracle stays close to the reference's own gradient (differs only where a branch flipped)
        assert rmsrel(oins[i].grad, T(g[f"din{i}"])) <= 2e-2, i
    params, oparams = dict(blk.named_parameters()), dict(oblk.named_parameters())
    for k, p in params.items():                           # EVERY parameter gradient of the block
        assert relerr(p.grad.cpu(), oparams[k].grad) <= TOL_BLOCK_DW, k
    _set_drop(blk, 0.3)
    _restore(hipmod)
    _restore(oracle)


def test_g8_hrnet_eval(golden, hipmod, prec):
    g = golden("g8_hrnet_eval")
    _restore(hipmod)
    hipmod.eval()
    with torch.no_grad():
        feat = hipmod.encoder(T(g["x"]).to(DEV), return_feature_maps=True)[0]
    assert feat.shape == (1, 960, 80, 80)
    feat = feat.cpu()
    scale = max(1.0, np.abs(g["crop"]).max())
    assert np.abs(feat[0, :, 32:48, 32:48].numpy() - g["crop"]).max() <= 1e-4 * scale
    assert np.abs(feat.mean(dim=(0, 2, 3)).numpy() - g["chan_mean"]).max() <= 1e-4 * scale


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_g9_c1(golden, hipmod, oracle, mode, prec):
    g = golden(f"g9_c1_{mode}")
    _restore(hipmod)
    _restore(oracle)
    hipmod.decoder.train(mode == "train")
    oracle.decoder.train(mode == "train")
    gg = torch.Generator().manual_seed(int(g["seed"]))
    f9 = torch.randn(2, 960, 80, 80, generator=gg) * 0.5
    fd = f9.to(DEV).requires_grad_(True)
    with traced() as tr:
        pred = hipmod.decoder([fd])
    cot = torch.randn(pred.shape, generator=gg) * 0.01
    hipmod.decoder.zero_grad()
    pred.backward(cot.to(DEV))
    p = pred.detach().cpu()
    assert np.abs(p[:, :50, 0, 0].numpy() - g["pred_ch0"]).max() <= 2e-5          # the reference's own outputs
    assert np.abs(p[:, 50].numpy() - g["pred_last"]).max() <= 2e-5
    fr = f9.clone().requires_grad_(True)
    with replay(hipmod, tr):
        opred = oracle.decoder([fr])
        oracle.decoder.zero_grad()
        (opred * cot).sum().backward()
    assert np.abs(opred[:, :50, 0, 0].detach().numpy() - g["pred_ch0"]).max() <= 2e-5
    assert relerr(fd.grad.cpu(), fr.grad) <= 2e-4
    assert rmsrel(fr.grad[:, ::60, 20:36, 20:36], T(g["dfeat_crop"])) <= 2e-2     # replayed oracle vs the reference's gradient
    po = dict(oracle.decoder.named_parameters())
    scale = float(hipmod.decoder.cls_net.layer2[0].conv1[0].weight.grad.abs().max())
    for k, q in hipmod.decoder.named_parameters():        # every decoder parameter
        if analytic_zero_grad(k, mode == "train"):
            assert float(q.grad.abs().max()) <= 1e-3 * scale, k
            continue
        assert relerr(q.grad.cpu(), po[k].grad) <= 5e-4, k
    _restore(hipmod)
    _restore(oracle)


def test_c1_classification_gradient_joins_the_mask_branch_epilogue(hipmod):
    """C1 reads `feat` twice.  The classification branch's gradient is handed (ops.StashGrad) to the bwd-data epilogue of the mask branch's
    3x3 conv instead of being added by a pass of its own: same dfeat and parameter gradients as with the two readers left to autograd, the
    fused launch really carries the addend, nothing is left stashed."""
    from fovealseg import modules as Mods
    _restore(hipmod)
    dec = hipmod.decoder
    dec.train(True)
    gg = torch.Generator().manual_seed(99)
    f9 = torch.randn(2, 960, 80, 80, generator=gg) * 0.5
    cot = None
    got = {}
    keep = Mods.C1_STASH
    real = fovealseg.hip.call
    try:
        for stash in (True, False):
            Mods.C1_STASH = stash
            ops.reset_step_state()
            addends = []

            def spy(name, *args, _real=real, _a=addends):
                if name == "fs_conv2d_bwd_data_bnsum" and args[-2] not in (None, 0):
                    _a.append(name)
                return _real(name, *args)
            fovealseg.hip.call = ops.hip.call = spy
            fd = f9.to(DEV).requires_grad_(True)
            pred = dec([fd])
            if cot is None:
                cot = (torch.randn(pred.shape, generator=gg) * 0.01).to(DEV)
            dec.zero_grad()
            pred.backward(cot)
            fovealseg.hip.call = ops.hip.call = real
            assert not ops.PENDING_RES
            assert (len(addends) == 1) == stash, (stash, addends)
            got[stash] = (fd.grad.cpu().clone(), {k: q.grad.detach().cpu().clone() for k, q in dec.named_parameters()})
    finally:
        fovealseg.hip.call = ops.hip.call = real
        Mods.C1_STASH = keep
    assert relerr(got[True][0], got[False][0]) <= 1e-6
    for k, v in got[False][1].items():
        # (weight gradients are split-K atomic sums: equal up to the order of the adds)
        assert relerr(got[True][1][k], v) <= 1e-5 or float(v.abs().max()) == 0.0, k
    _restore(hipmod)


class _InjectValue(torch.autograd.Function):
    """forward: the injected value; backward: gradient flows to the computed tensor."""

    @staticmethod
    def forward(ctx, computed, value):
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return g, None


def _feed(g):
    return {"img_data": T(g["x"]).to(DEV), "seg_label": T(g["y"]).to(DEV), "focus_point": T(g["focus"]).to(DEV),
            "cls_label": T(g["cls"]).to(DEV)}


def test_g11_end_to_end(golden, hipmod, oracle, prec):
    """End to end against the reference run.  The reference's own fp32 grid is 1.75e-5 from the fp64
    value of its formula; a 1.5e-5 perturbation of the grid flips 0.2 % of the truncated
    labels and moves the reference's OWN gradient norms by up to 21 % (measured with the oracle), so
    (a) the free-running path gets a label-flip / scalar budget, and (b) tight gradient parity is
    checked with the reference's grid injected (bit-identical labels and x_sampled)."""
    try:
        g = golden("g11_e2e_eval")
        _restore(hipmod)
        hipmod.eval()
        feed = _feed(g)
        with torch.no_grad():
            outs = hipmod(feed, is_inference=True)
        assert len(outs) == 6
        got = np.array([float(o) for o in outs])
        flips = float((feed["seg_label"].cpu().numpy() != g["label"]).mean())
        assert flips <= 3e-3, flips
        assert np.abs(got - g["outs"]).max() <= 2e-3, (got, g["outs"])

        g = golden("g11_e2e_train_p0")
        _set_drop(hipmod, 0.0)
        # (a) free running
        _restore(hipmod)
        hipmod.train()
        feed = _feed(g)
        hipmod.zero_grad()
        loss, acc, edge = hipmod(feed)
        loss.mean().backward()
        got = np.array([float(loss), float(acc), float(edge)])
        assert np.abs(got - g["outs"]).max() <= 2e-3, (got, g["outs"])
        assert float((feed["seg_label"].cpu().numpy() != g["label"]).mean()) <= 3e-3
        assert abs(float(edge) - float(g["outs"][2])) <= 1e-5
        grid_free = hipmod.create_grid(hipmod.saliency(feed["img_data"], feed["focus_point"])[0]).detach().cpu().numpy()
        assert np.abs(grid_free - g["grid"]).max() <= 3e-5
        # (b) reference grid injected on both sides; the oracle replays the device's activation branches
        _restore(hipmod)
        _restore(oracle)
        hipmod.train()
        oracle.train()
        ref_grid = T(g["grid"]).to(DEV)
        orig = hipmod.create_grid
        hipmod.create_grid = lambda xs: _InjectValue.apply(orig(xs), ref_grid)
        try:
            feed = _feed(g)
            hipmod.zero_grad()
            with traced() as tr:
                loss, acc, edge = hipmod(feed)
            loss.mean().backward()
        finally:
            del hipmod.create_grid
        assert np.array_equal(feed["seg_label"].cpu().numpy(), g["label"])          # bit-exact label map
        got = np.array([float(loss), float(acc), float(edge)])
        assert np.abs(got - g["outs"]).max() <= 1e-4, (got, g["outs"])
        oorig = oracle.grid_from_saliency
        oracle.grid_from_saliency = lambda xs: _InjectValue.apply(oorig(xs), T(g["grid"]))
        try:
            ofeed = {"img_data": T(g["x"]), "seg_label": T(g["y"]).clone(), "focus_point": T(g["focus"]), "cls_label": T(g["cls"])}
            oracle.zero_grad()
            with replay(hipmod, tr):
                oloss, oacc, oedge = oracle(ofeed, drop_fn=lambda n, t: t)
                oloss.backward()
        finally:
            del oracle.grid_from_saliency
        assert abs(float(oloss) - float(g["outs"][0])) <= 1e-4
        params, oparams = dict(hipmod.named_parameters()), dict(oracle.named_parameters())
        worst = {}
        for n, q in params.items():
            if q.grad is None or n not in oparams or oparams[n].grad is None or analytic_zero_grad(n, True):
                continue
            e = rmsrel(q.grad.cpu(), oparams[n].grad)
            grp = n.split(".")[0]
            if e > worst.get(grp, ("", 0.0))[1]:
                worst[grp] = (n, e)
        print("g11 worst rms-relative gradient error per net", prec, worst)
        # EVERY parameter-gradient tensor of all four nets, one bound for all three modes (measured: 0.9e-4 .. 2.4e-4).  With the
        # same grid on both sides even the saliency side (through the clamp mask of create_grid) is tight.
        for grp in ("encoder", "decoder", "localization", "net_compress"):
            assert worst[grp][1] <= TOL_E2E_GRAD, (prec, worst[grp])
        for n, ref in zip(g["gn_names"], g["gn"]):                                 # replayed oracle vs the reference's own norms
            gn = float(oparams[str(n)].grad.norm())
            tol = 5e-2 if (str(n).startswith("localization") or str(n).startswith("net_compress")) else 2e-2
            assert abs(gn - float(ref)) <= tol * max(abs(float(ref)), 1e-6), (n, gn, ref)
    finally:
        _set_drop(hipmod, 0.3)
        _restore(hipmod)
        _restore(oracle)


def test_dropout_replay_basic_block(hipmod, oracle, prec):
    """Train-mode BasicBlock with Dropout(0.3): the kernel's hash mask AND its activation branches replayed in the oracle."""
    _restore(hipmod)
    _restore(oracle)
    path = "stage3.1.branches.1.2"
    blk = _sub(hipmod.encoder, path).train()
    oblk = _sub(oracle.encoder, path).train()
    ops.DropoutState.seed, ops.DropoutState.step = 5, 17
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 128, 10, 10, generator=g)

    def drop_fn(name, t):
        B, C, H, W = t.shape
        key = ops.DropoutState.key(ops.layer_id_from_name(name))
        keep = O.dropout_keep_mask_nhwc(t.numel(), key, 0.3).reshape(B, H, W, C)
        return t * torch.from_numpy(keep).permute(0, 3, 1, 2).float() * np.float32(1.0 / 0.7)
    xd = nhwc(x).requires_grad_(True)
    with traced() as tr:
        out = blk(xd)
    cot = torch.randn(2, 128, 10, 10, generator=g)
    blk.zero_grad()
    out.backward(nhwc(cot))
    xr = x.clone().requires_grad_(True)
    with replay(hipmod, tr):
        ref = oblk(xr, O._Ctx(True, drop_fn), path)
        oblk.zero_grad()
        ref.backward(cot)
    assert relerr(nchw(out), ref.detach()) <= TOL_BLOCK_OUT
    assert relerr(nchw(xd.grad), xr.grad) <= TOL_BLOCK_DIN
    po = dict(oblk.named_parameters())
    for k, q in blk.named_parameters():
        assert relerr(q.grad.cpu(), po[k].grad) <= TOL_BLOCK_DW, k
    frac = float((nchw(out) == 0).float().mean())
    assert 0.0 < frac < 1.0


# ------------------------------------------------------------------------------------------------
# full depth at the bench shape: HRNetV2 + C1 + Dice/Focal, train mode, B = 64 at 80x80, every precision mode
# ------------------------------------------------------------------------------------------------
FULL_DEPTH_LOGIT_TOL = 1e-4       # north_star: "fp32 logits within 1e-4"
FULL_DEPTH_GRAD_TOL = 4e-4        # rms-relative per parameter-gradient tensor, the same for f32 / bf16x3 / f16x2 (measured worst: 1.0e-4 / 1.3e-4 / 1.1e-4)


# (mode, spatial size): all three modes at the bench shape, B = 64 at 80x80 (~55 s each, most of it the CPU oracle's forward +
# backward on the box's 16 cores), plus a ragged 40x40 case for the default mode (5x5 maps in the lowest branch, global pooling)
@pytest.mark.parametrize("mode,hw", [("f16x2", 80), ("bf16x3", 80), ("f32", 80), ("f16x2", 40)])
def test_full_depth_b64_modes(hipmod, oracle, mode, hw):
    """VERDICT r1 "next" #1: the claim that the split-precision conv modes sit at the fp32 error level, as a test at FULL DEPTH
    and at the BENCH batch: encoder (HRNetV2-nodownsp, ~300 conv+BN layers) -> C1 -> Dice+Focal, train mode (batch
    statistics; Dropout at p = 0), B = 64, 80x80, forward and backward, against the CPU oracle with the device's activation
    branches replayed: logits <= 1e-4, and EVERY parameter-gradient tensor of encoder and decoder within one rms-relative
    bound that does not depend on the mode."""
    B = 64
    gen = torch.Generator().manual_seed(64)
    x = torch.rand(B, 3, hw, hw, generator=gen)
    cls = torch.randint(0, 50, (B, 1, 1), generator=gen)
    ii = torch.arange(hw, dtype=torch.float32)
    c = torch.rand(B, 2, generator=gen) * (0.75 * hw) + 0.125 * hw
    disc = ((ii[None, :, None] - c[:, 0, None, None]) ** 2 + (ii[None, None, :] - c[:, 1, None, None]) ** 2) <= (0.19 * hw) ** 2
    gt = torch.where(disc, cls.expand(B, hw, hw), torch.full((B, hw, hw), 50))
    fovealseg.hip.set_conv_precision(mode)
    try:
        _restore(hipmod)
        _restore(oracle)
        hipmod.train()
        oracle.train()
        _set_drop(hipmod, 0.0)
        hipmod.zero_grad()
        with traced() as tr:
            feat = hipmod.encoder.forward_nhwc(nhwc(x))
            pred = hipmod.decoder.forward_nhwc(feat)
        loss = ops.SegLoss.apply(pred, gt.to(DEV), 5.0)[0]
        loss.backward()
        torch.cuda.synchronize()
        rp = replay(hipmod, tr)
        del tr, feat
        oracle.zero_grad()
        with rp:
            opred = oracle.decoder(oracle.encoder(x, return_feature_maps=True, drop_fn=lambda n, t: t))
            oloss = O.dice_loss_multiclass(opred, gt) + O.focal_loss(opred, gt)
            oloss.backward()
        pd = pred.detach().cpu()
        err_logit = float((pd - opred.detach()).abs().max() / max(1.0, float(opred.detach().abs().max())))
        # ... and against an UN-replayed oracle forward (VERDICT r2 #5a): the replayed pass zeroes whatever the device zeroed, so a
        # device activation that wrongly killed a positive pre-activation would be reproduced by it; the forward is continuous in
        # its inputs, so the plain oracle must agree to the same tolerance without any help
        with torch.no_grad():
            opred_free = oracle.decoder(oracle.encoder(x, return_feature_maps=True, drop_fn=lambda n, t: t))
        err_free = float((pd - opred_free).abs().max() / max(1.0, float(opred_free.abs().max())))
        assert err_free <= FULL_DEPTH_LOGIT_TOL, ("un-replayed oracle", err_free)
        assert abs(float(loss) - float(oloss)) <= 1e-5 * max(1.0, abs(float(oloss)))
        po = dict(oracle.named_parameters())
        errs = {}
        for n, q in hipmod.named_parameters():
            if n.startswith("encoder.") or n.startswith("decoder."):
                if ".cls_net.layer" in n and n.endswith(".0.bias"):
                    # a conv bias in front of a batch-statistics BatchNorm: analytically zero gradient, both sides hold rounding noise
                    assert float(q.grad.norm()) <= 1e-4 * float(dict(hipmod.named_parameters())[n[:-4] + "weight"].grad.norm()), n
                    continue
                errs[n] = rmsrel(q.grad.cpu(), po[n].grad)
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
        med = float(np.median(list(errs.values())))
        print(f"full depth B=64 {hw}x{hw} {mode}: logits {err_logit:.2e} (un-replayed oracle {err_free:.2e}); {len(errs)} gradient tensors, rms-rel median {med:.2e}, worst {worst}")
        assert err_logit <= FULL_DEPTH_LOGIT_TOL, err_logit
        assert worst[0][1] <= FULL_DEPTH_GRAD_TOL, worst
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())
        _set_drop(hipmod, 0.3)
        hipmod.zero_grad(set_to_none=True)
        _restore(hipmod)
        _restore(oracle)


# ------------------------------------------------------------------------------------------------
# BASELINE configs[1] at FULL size, end to end against the oracle (VERDICT r2 #5b): B = 64, 1024x1024 -> 80x80, train mode, bf16x3
# ------------------------------------------------------------------------------------------------
def test_config1_full_size_end_to_end(hipmod, oracle):
    """The front end at its real size (the G2/G5/G6 goldens stop at 640x640): one training forward + backward of the whole module on
    the bench batch (B = 64, 1024x1024, Dropout at p = 0 so both sides see the same network).
    Free running: the device grid is within 3e-5 of the oracle's (the reference's own fp32 grid is 1.75e-5 from fp64) and at
    most 0.3 % of the truncated labels differ.  With the ORACLE's grid injected on the device side: sampled label maps bit-exact
    (int64), x_sampled bit-exact (fp32), (loss, acc, edge) within 1e-4, every saliency / compress gradient tensor within 1e-3
    rms-relative (encoder / decoder gradients at this batch are test_full_depth_b64_modes' subject)."""
    from fovealseg import train as Tr
    B, H = 64, 1024
    X, Fp, Y, cls = Tr.synthetic_batch(B, H, H, seed=1, device="cpu")
    fovealseg.hip.set_conv_precision("bf16x3")
    try:
        _restore(hipmod)
        _restore(oracle)
        hipmod.train()
        oracle.train()
        _set_drop(hipmod, 0.0)
        dev_feed = lambda: {"img_data": X.to(DEV), "seg_label": Y.to(DEV), "focus_point": Fp.to(DEV), "cls_label": cls.to(DEV)}  # noqa: E731
        # ---- oracle, free running (its grid is the one injected below); activation branches replayed from the injected device run ----
        # (1) device free running: grid and label flips against the oracle's own free-running front end
        with torch.no_grad():
            oxs, _ = oracle.saliency(X, Fp)
            ogrid = oracle.grid_from_saliency(oxs).contiguous()
            olabel = F.grid_sample(Y.float(), ogrid, align_corners=False).squeeze(1).long()
            ox_s = F.grid_sample(X, ogrid, align_corners=False)
            dxs, _ = hipmod.saliency(X.to(DEV), Fp.to(DEV))
            dgrid = hipmod.create_grid(dxs)
            assert float((dgrid.cpu() - ogrid).abs().max()) <= 3e-5
            dlabel = ops.grid_sample_label(Y.to(DEV), dgrid)
            flips = float((dlabel.cpu() != olabel).float().mean())
            assert flips <= 3e-3, flips
        # (2) oracle grid injected on the device side
        ref_grid = ogrid.contiguous().to(DEV)
        orig = hipmod.create_grid
        hipmod.create_grid = lambda xs: _InjectValue.apply(orig(xs), ref_grid)
        try:
            feed = dev_feed()
            hipmod.zero_grad()
            with traced() as tr:
                loss, acc, edge = hipmod(feed)
            loss.mean().backward()
            x_s = ops.GridSample.apply(X.to(DEV), ref_grid)
        finally:
            del hipmod.create_grid
        torch.cuda.synchronize()
        assert torch.equal(feed["seg_label"].cpu(), olabel)                       # bit-exact int64 label maps at full size
        assert torch.equal(nchw(x_s), ox_s)                                       # bit-exact fp32 foveated image at full size
        rp = replay(hipmod, tr)
        del tr
        ofeed = {"img_data": X, "seg_label": Y.clone(), "focus_point": Fp, "cls_label": cls}
        oracle.zero_grad()
        with rp:
            oloss, oacc, oedge = oracle(ofeed, drop_fn=lambda n, t: t)
            oloss.backward()
        got = np.array([float(loss), float(acc), float(edge)])
        want = np.array([float(oloss), float(oacc), float(oedge)])
        assert np.abs(got - want).max() <= 1e-4, (got, want)
        params, oparams = dict(hipmod.named_parameters()), dict(oracle.named_parameters())
        worst = ("", 0.0)
        for n, q in params.items():
            if not (n.startswith("localization.") or n.startswith("net_compress.")):
                continue
            if q.grad is None or oparams[n].grad is None or analytic_zero_grad(n, True):
                continue
            e = rmsrel(q.grad.cpu(), oparams[n].grad)
            if e > worst[1]:
                worst = (n, e)
        print(f"configs[1] full size: grid err {float((dgrid.cpu() - ogrid).abs().max()):.2e}, label flips {flips:.2e}, scalars {got} vs {want}, worst saliency-side gradient {worst}")
        assert worst[1] <= 1e-3, worst
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())
        _set_drop(hipmod, 0.3)
        hipmod.zero_grad(set_to_none=True)
        _restore(hipmod)
        _restore(oracle)


# ------------------------------------------------------------------------------------------------
# BASELINE config-2 sizes: size-independent properties (B=64, 1024x1024 -> 80x80)
# ------------------------------------------------------------------------------------------------
def test_full_size_properties():
    B, H = 64, 1024
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 3, H, H, generator=g).to(DEV)
    focus = (torch.rand(B, 2, generator=g) * 0.8 + 0.1).to(DEV)
    # (1) an identity grid at the low-res pixel centres reproduces F.interpolate-free bilinear taps:
    #     sampling a constant image returns the constant exactly inside, and linearity holds
    ys, xs_ = torch.meshgrid(torch.linspace(-0.95, 0.95, 80), torch.linspace(-0.95, 0.95, 80), indexing="ij")
    grid = torch.stack((xs_, ys), -1)[None].repeat(B, 1, 1, 1).contiguous().to(DEV)
    ones = torch.ones(B, 3, H, H, device=DEV)
    s1 = ops.GridSample.apply(ones, grid)
    assert float((s1 - 1).abs().max()) <= 2e-7
    a = ops.GridSample.apply(x, grid)
    b2 = ops.GridSample.apply(x * 2, grid)
    assert torch.equal(a * 2, b2)                                   # exact: power-of-two scaling
    # (2) label map of an all-ones mask is 1 wherever all four taps are in range
    lab = ops.grid_sample_label(torch.ones(B, 1, H, H, device=DEV), grid)
    assert int(lab.min()) == 0 or int(lab.min()) == 1
    assert int(lab.sum()) >= int(0.9 * lab.numel())
    # (3) low-res input: gaze channel is zero at the gaze pixel and both copies are identical
    xl = ops.gaze_lowres(x, focus, 80, 80)
    assert torch.equal(xl[..., 3], xl[..., 4])
    assert float(xl[..., :3].min()) >= 0 and float(xl[..., :3].max()) <= 1
    # (4) area pool of a constant is the constant; softmax saliency sums to one per image
    ap = ops.area_pool(torch.full((B, 1, H, H), 0.25, device=DEV), 80, 80)
    assert float((ap - 0.25).abs().max()) <= 1e-7
    # (5) uniform saliency -> (near-)uniform grid, symmetric about the centre
    uni = torch.full((B, 1, 80, 80), 1.0 / 6400, device=DEV)
    g1d = torch.from_numpy(O.gaussian_1d(91, 45)).to(DEV)
    gr = ops.GaussGrid.apply(uni, g1d, 45)
    assert float((gr[:, :, :, 0] + gr[:, :, :, 0].flip(2)).abs().max()) <= 1e-6
    assert float((gr[:, :, :, 1] + gr[:, :, :, 1].flip(1)).abs().max()) <= 1e-6


# ------------------------------------------------------------------------------------------------
# optimiser + full training step
# ------------------------------------------------------------------------------------------------
def test_flat_adam_matches_torch_adam():
    from fovealseg import train
    g = torch.Generator().manual_seed(21)
    shapes = [(8, 4, 3, 3), (5,), (7, 3)]
    cpu = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    dev = []
    for p in cpu:
        if p.dim() == 4:
            t = ops.new_rsck_weight(*p.shape, device=DEV)
            t.copy_(p.detach())
        else:
            t = p.detach().clone().to(DEV)
        dev.append(torch.nn.Parameter(t))
    ref = torch.optim.Adam(cpu, lr=1e-2, weight_decay=1e-4)
    opt = train.FlatAdam(dev, lr=1e-2, weight_decay=1e-4, lr_mult=1.0, zoom=False)
    for step in range(3):
        opt.zero_grad()
        for pc, pd in zip(cpu, dev):
            gr = torch.randn(pc.shape, generator=g)
            pc.grad = gr.clone()
            pd.grad.copy_(gr)
        ref.step()
        opt.step()
    for pc, pd in zip(cpu, dev):
        assert pd.permute(2, 3, 1, 0).is_contiguous() if pd.dim() == 4 else True
        assert np.abs(pd.detach().cpu().numpy() - pc.detach().numpy()).max() <= 2e-6


def test_train_and_eval_step_end_to_end():
    """Two optimisation steps + one eval step through fovealseg.train on a small batch: finite losses,
    parameters move, BN running stats update, eval returns the reference's 6-tuple."""
    from fovealseg import train
    cfg = fovealseg.lvis50_cfg()
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    opts = train.create_optimizers(nets, cfg)
    try:
        batch = train.synthetic_batch(2, 128, 128, seed=5, device=DEV)
        w0 = module.encoder.conv1.weight.detach().clone()
        rm0 = module.encoder.bn1.running_mean.clone()
        losses = []
        for it in range(2):
            out = train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=it)
            assert len(out) == 3
            losses.append(float(out[0].detach()))
        assert all(np.isfinite(losses))
        assert opts[0].param_groups[0]["lr"] == 0.001 * 0.1
        assert float((module.encoder.conv1.weight.detach() - w0).abs().max()) > 0
        assert float((module.encoder.bn1.running_mean - rm0).abs().max()) > 0
        assert int(module.state_dict()["encoder.bn1.num_batches_tracked"]) == 2
        g = module.encoder.conv1.weight.grad
        assert g.data_ptr() >= opts[0].flat.grad.data_ptr() and bool(torch.isfinite(g).all())
        module.eval()
        outs = train.eval_step(module, batch)
        assert len(outs) == 6 and all(np.isfinite(float(o)) for o in outs)
    finally:
        pass          # (round 1 reset a process-global here; the direct-gradient decision is per parameter now)


# ------------------------------------------------------------------------------------------------
# DeepLab encoder plugin -- against this repo's oracle restatement ("parity unpinned"
# w.r.t. torchvision, which is absent)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_deeplab_encoder_vs_oracle(mode, prec):
    import deeplab_oracle as DO
    from fovealseg import deeplab as D
    o = DO.OracleDeepLab()
    fovealseg.weights.apply_name_keyed_init(o)
    m = D.deeplab()
    m.load_state_dict(o.state_dict(), strict=True)
    m.to(DEV)
    o.train(mode == "train")
    m.train(mode == "train")
    ops.DropoutState.seed, ops.DropoutState.step = 3, 11
    g = torch.Generator().manual_seed(23)
    x = torch.rand(4, 3, 80, 80, generator=g)

    def drop_fn(t):                       # replay the kernel's hash mask (Dropout(0.5) of the ASPP projection)
        if mode != "train":
            return t
        B, C, H, W = t.shape
        key = ops.DropoutState.key(ops.layer_id_from_name("deeplab.classifier.0.project.3"))
        keep = O.dropout_keep_mask_nhwc(t.numel(), key, 0.5).reshape(B, H, W, C)
        return t * torch.from_numpy(keep).permute(0, 3, 1, 2).float() * 2.0
    O.assign_paths(o)
    xd = x.to(DEV).requires_grad_(True)
    with traced() as tr:
        out = m(xd)[0]
    assert out.shape == (4, 960, 80, 80)
    cot = torch.randn(out.shape, generator=g) * 0.01
    m.zero_grad()
    out.backward(cot.to(DEV))
    xr = x.clone().requires_grad_(True)
    with replay(m, tr):                   # the oracle takes the activation branches the device took
        ref = o(xr, drop_fn=drop_fn)[0]
        o.zero_grad()
        ref.backward(cot)
    # train mode at B=4 normalises the ASPP image-pooling branch over FOUR samples (and 10x10 maps over 400):
    # batch statistics of so few values amplify rounding differences (the oracle's own fp32 and fp64 runs differ by 7e-4 in
    # the output in train mode), hence the looser train tolerances; the pooling-branch conv gradient (BatchNorm over 4
    # values, analytically near-cancelling) is checked in eval only; tight train-mode checks per block: test_deeplab_blocks_train
    tol_out, tol_grad = (1e-4, 2e-3) if mode == "eval" else (2e-3, DEEPLAB_TRAIN_GRAD_TOL)
    assert relerr(out.detach().cpu(), ref.detach()) <= tol_out
    print("deeplab", mode, prec, "dx", relerr(xd.grad.cpu(), xr.grad))
    assert relerr(xd.grad.cpu(), xr.grad) <= tol_grad
    po, pm = dict(o.named_parameters()), dict(m.named_parameters())
    for k in ("deeplab.backbone.conv1.weight", "deeplab.backbone.layer3.7.conv2.weight", "deeplab.classifier.0.convs.2.0.weight",
              "deeplab.classifier.0.convs.4.1.weight", "deeplab.classifier.4.bias", "deeplab.classifier.1.weight",
              "deeplab.backbone.layer4.2.bn3.weight"):
        if mode == "train" and "convs.4" in k:
            continue
        print("deeplab", mode, prec, k, relerr(pm[k].grad.cpu(), po[k].grad))
        assert relerr(pm[k].grad.cpu(), po[k].grad) <= tol_grad, k


DEEPLAB_TRAIN_GRAD_TOL = 5e-3     # was 0.25 before the activation replay; measured 3e-4 .. 1.3e-3 in all modes


def test_deeplab_blocks_train(prec):
    """Train-mode (batch statistics, Dropout(0.5) replayed) parity of the DeepLab-specific blocks in isolation:
    a dilated bottleneck and the ASPP head, where the comparison is well conditioned."""
    import deeplab_oracle as DO
    from fovealseg import deeplab as D
    o = DO.OracleDeepLab()
    fovealseg.weights.apply_name_keyed_init(o)
    m = D.deeplab()
    m.load_state_dict(o.state_dict(), strict=True)
    m.to(DEV)
    o.train()
    m.train()
    g = torch.Generator().manual_seed(29)
    # dilated bottleneck (layer3[7]: dilation 2)
    O.assign_paths(o)
    x = torch.randn(2, 1024, 10, 10, generator=g)
    ob, mb = o.deeplab.backbone.layer3[7], m.deeplab.backbone.layer3[7]
    xd = nhwc(x).requires_grad_(True)
    with traced() as tr:
        out = mb(xd)
    cot = torch.randn(2, 1024, 10, 10, generator=g)
    mb.zero_grad()
    out.backward(nhwc(cot))
    xr = x.clone().requires_grad_(True)
    with replay(m, tr):
        ref = ob(xr)
        ob.zero_grad()
        ref.backward(cot)
    assert relerr(nchw(out), ref.detach()) <= 2e-5
    assert relerr(nchw(xd.grad), xr.grad) <= 2e-4
    pob = dict(ob.named_parameters())
    for k, q in mb.named_parameters():
        assert relerr(q.grad.cpu(), pob[k].grad) <= 5e-4, k
    # ASPP head; the image-pooling BN normalises over B values only -> kept in eval mode on both sides
    oh, mh = o.deeplab.classifier, m.deeplab.classifier
    oh[0].convs[4][2].eval()
    mh[0].convs[4][2].eval()
    ops.DropoutState.seed, ops.DropoutState.step = 4, 2
    f = torch.randn(4, 2048, 10, 10, generator=g) * 0.5
    fr = f.clone().requires_grad_(True)

    def drop_fn(t):
        B, C, H, W = t.shape
        key = ops.DropoutState.key(ops.layer_id_from_name("deeplab.classifier.0.project.3"))
        keep = O.dropout_keep_mask_nhwc(t.numel(), key, 0.5).reshape(B, H, W, C)
        return t * torch.from_numpy(keep).permute(0, 3, 1, 2).float() * 2.0
    fd = nhwc(f).requires_grad_(True)
    with traced() as tr:
        out = mh(fd)
    cot = torch.randn(4, out.shape[3], 10, 10, generator=g) * 0.1
    mh.zero_grad()
    out.backward(nhwc(cot))
    with replay(m, tr):
        ref = oh[4](oh[3](oh[2](oh[1](oh[0](fr, drop_fn)))))
        oh.zero_grad()
        ref.backward(cot)
    assert relerr(nchw(out), ref.detach()) <= 5e-5
    assert relerr(nchw(fd.grad), fr.grad) <= 1e-3
    # ("1.bias" is omitted: a conv bias in front of a batch-stat BN has an analytically zero gradient)
    for k in ("0.convs.2.0.weight", "0.convs.4.1.weight", "0.project.0.weight", "1.weight", "4.weight", "4.bias", "2.weight"):
        po, pm = dict(oh.named_parameters())[k], dict(mh.named_parameters())[k]
        assert relerr(pm.grad.cpu(), po.grad) <= 1e-3, k


@pytest.mark.parametrize("train_mode", [True, False])
def test_deeplab_atrous_runs_in_the_phase_domain(train_mode):
    """Round 5: layer3's 22 dilation-2 blocks and layer4's first block run as ONE stretch in the phase domain of their dilation (two re-ordering
    copies per pass instead of two per atrous 3x3): same network output and gradients as with every 3x3 re-ordering around itself (the
    route the oracle tests pin: they trace activations, which switches the chained form off), to the rounding of BatchNorm sums taken in
    another order; the dilation-4 blocks (10 is not a multiple of 4) leave the domain."""
    from fovealseg import deeplab as D
    from fovealseg import modules as Mods
    torch.manual_seed(3)
    m = D.deeplab().to(DEV)
    m.train(train_mode)
    bb = m.deeplab.backbone
    x0 = torch.randn(2, 80, 80, 3, device=DEV)
    cot = None

    def run(chained):
        nonlocal cot
        saved = D.PHASE_DOMAIN
        D.PHASE_DOMAIN = chained
        copies = []
        real_s2b, real_b2s = Mods._space_to_batch, Mods._batch_to_space
        Mods._space_to_batch = lambda t, d: (copies.append(("s2b", d)), real_s2b(t, d))[1]
        Mods._batch_to_space = lambda t, d: (copies.append(("b2s", d)), real_b2s(t, d))[1]
        try:
            bb.zero_grad()
            x = x0.clone().requires_grad_(True)
            out = bb(x)
            if cot is None:
                cot = torch.randn_like(out) * 0.1
            out.backward(cot)
            return out.detach(), x.grad, {k: p.grad.clone() for k, p in bb.named_parameters() if p.grad is not None}, copies
        finally:
            D.PHASE_DOMAIN = saved
            Mods._space_to_batch, Mods._batch_to_space = real_s2b, real_b2s
    o1, g1, p1, c1 = run(True)
    o0, g0, p0, c0 = run(False)
    assert o1.shape == o0.shape == (2, 10, 10, 2048)
    assert c1 == [("s2b", 2), ("b2s", 2)], c1                     # one stretch: layer3[1:] + layer4[0]
    assert len(c0) == 2 * 23                                        # 22 + 1 atrous 3x3 layers, each with its own pair
    if train_mode:
        # batch statistics over 200 values per channel, 70 BatchNorm layers deep: the other summation order of the phase-domain slabs shows
        # as 2.4e-5 in the output, and a handful of ReLU decisions that flip on it as 1.4e-2 in the input gradient (measured) -- the gradients
        # of the whole stack are compared in eval mode, and in train mode on a stretch of three blocks below, where the comparison is
        # well conditioned
        assert relerr(o1, o0) <= 1e-4
        blocks = list(bb.layer3)[1:4]
        xs = torch.randn(2, 10, 10, 1024, device=DEV)
        cs = torch.randn(2, 10, 10, 1024, device=DEV)

        def stretch(chained):
            for b_ in blocks:
                b_.zero_grad()
            x = xs.clone().requires_grad_(True)
            h = Mods._space_to_batch(x, 2) if chained else x
            for b_ in blocks:
                h = b_(h, phase=2 if chained else 1)
            out = Mods._batch_to_space(h, 2) if chained else h
            out.backward(cs)
            return out.detach(), x.grad, [p.grad.clone() for b_ in blocks for p in b_.parameters()]
        so1, sg1, sp1 = stretch(True)
        so0, sg0, sp0 = stretch(False)
        assert relerr(so1, so0) <= 5e-6 and relerr(sg1, sg0) <= 5e-5
        for a, b in zip(sp1, sp0):
            assert float((a - b).abs().max()) <= 2e-4 * (float(b.abs().max()) + 1e-30)
        return
    assert relerr(o1, o0) <= 1e-5 and relerr(g1, g0) <= 2e-4
    for k in p0:
        assert relerr(p1[k], p0[k]) <= 2e-4, k


def test_deeplab_through_module_surface():
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder = "deeplab"
    from fovealseg import train
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    batch = train.synthetic_batch(2, 128, 128, seed=9, device=DEV)
    feed = {"img_data": batch[0], "seg_label": batch[2], "focus_point": batch[1], "cls_label": batch[3]}
    loss, acc, edge = module(feed)
    loss.mean().backward()
    assert np.isfinite(float(loss.detach())) and module.encoder.deeplab.backbone.conv1.weight.grad is not None


# ------------------------------------------------------------------------------------------------
# SegFormer encoder plugin: kernels vs torch, encoder vs oracle / transformers-5.15 golden
# ------------------------------------------------------------------------------------------------
def test_layernorm_gelu_dwconv_droppath():
    g = torch.Generator().manual_seed(41)
    # 16-lanes-per-row kernels for C <= 320 (one to five float4s per lane; 132 = ragged last float4 column), a wave per row above;
    # 111 rows = a ragged last group of 16, 40 003 rows = several 16-row rounds per workgroup in the backward
    for C, lead in ((64, (3, 37)), (128, (3, 37)), (132, (3, 37)), (320, (3, 37)), (512, (3, 37)), (2048, (3, 37)), (64, (1, 40003)), (320, (1, 5001))):
        x = torch.randn(*lead, C, generator=g)
        w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref = F.layer_norm(xr, (C,), wr, br, 1e-6)
        cot = torch.randn(ref.shape, generator=g)
        ref.backward(cot)
        xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        out = ops.LayerNorm.apply(xd, wd, bd, 1e-6)
        out.backward(cot.to(DEV))
        assert relerr(out.detach().cpu(), ref.detach()) <= 1e-5
        assert relerr(xd.grad.cpu(), xr.grad) <= 1e-4
        assert relerr(wd.grad.cpu(), wr.grad) <= 1e-4 and relerr(bd.grad.cpu(), br.grad) <= 1e-4
        # the pre-norm fan-out as one node (round 5): y = LN(x) and x itself; the residual's gradient is added inside the LayerNorm backward
        # pass.  Same outputs bit for bit; dx = the separate node's dx + the skip gradient, the one rounding of that add apart
        cot2 = torch.randn(ref.shape, generator=g).to(DEV)
        xf, wf, bf = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        yf, xskip = ops.LayerNormFan.apply(xf, wf, bf, 1e-6)
        assert torch.equal(yf, out) and torch.equal(xskip, xf)
        torch.autograd.backward([yf, xskip], [cot.to(DEV), cot2])
        assert torch.equal(xf.grad, xd.grad + cot2)
        assert torch.equal(wf.grad, wd.grad) and torch.equal(bf.grad, bd.grad)
        xg = x.to(DEV).requires_grad_(True)          # only the skip output used / only the LayerNorm output used
        ops.LayerNormFan.apply(xg, wf, bf, 1e-6)[1].backward(cot2)
        assert torch.equal(xg.grad, cot2)
        xh = x.to(DEV).requires_grad_(True)
        ops.LayerNormFan.apply(xh, wf, bf, 1e-6)[0].backward(cot.to(DEV))
        assert torch.equal(xh.grad, xd.grad)
    x = torch.randn(2, 9, 7, 256, generator=g) * 2
    xr = x.clone().requires_grad_(True)
    ref = F.gelu(xr)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    xd = x.to(DEV).requires_grad_(True)
    out = ops.Gelu.apply(xd)
    out.backward(cot.to(DEV))
    assert relerr(out.detach().cpu(), ref.detach()) <= 1e-6 and relerr(xd.grad.cpu(), xr.grad) <= 1e-5
    # GELU + the Dropout behind it in one pass each way: bit-identical to the two separate ops (same mask, same arithmetic order)
    key = ops.layer_key(11, 5)
    xa, xb = x.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    ya = ops.GeluDropout.apply(xa, 0.3, key)
    yb = ops.Dropout.apply(ops.Gelu.apply(xb), 0.3, key)
    ya.backward(cot.to(DEV))
    yb.backward(cot.to(DEV))
    assert torch.equal(ya.detach(), yb.detach())
    assert relerr(xa.grad, xb.grad) <= 1e-7
    keep = torch.from_numpy(O.dropout_keep_mask_nhwc(x.numel(), key, 0.3)).view(x.shape)
    # (zero-ness of gelu(x) from the DEVICE gelu: at x ~ -5.5 the last ulp of erff decides between -0 and -1.6e-7, CPU and device differ there)
    assert torch.equal(ya.detach().cpu() != 0, keep & (ops.Gelu.apply(x.to(DEV)).cpu() != 0))
    # depthwise 3x3
    C = 256
    xi = torch.randn(2, C, 11, 13, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(C, generator=g) * 0.1
    xr, wr, br = xi.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, 1, 1, 1, C)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    xd, wd, bd = nhwc(xi).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    out = ops.DwConv3.apply(xd, wd, bd)
    out.backward(nhwc(cot))
    assert relerr(nchw(out), ref.detach()) <= 1e-5 and relerr(nchw(xd.grad), xr.grad) <= 1e-5
    assert relerr(wd.grad.cpu(), wr.grad) <= 1e-4 and relerr(bd.grad.cpu(), br.grad) <= 1e-4
    # (round 5: the bias gradient comes out of the weight-gradient launches; without a bias the nine-plane form runs: same dw, bit for bit)
    x2, w2 = nhwc(xi).requires_grad_(True), w.to(DEV).requires_grad_(True)
    ops.DwConv3.apply(x2, w2, None).backward(nhwc(cot))
    assert torch.equal(w2.grad, wd.grad) and torch.equal(x2.grad, xd.grad)
    # residual + DropPath replay
    key = ops.layer_key(2, 5)
    a, y = torch.randn(6, 10, 64, generator=g), torch.randn(6, 10, 64, generator=g)
    keep = torch.from_numpy(O.dropout_keep_mask_nhwc(6, key, 0.25)).float().view(6, 1, 1)
    ad, yd = a.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
    out = ops.ResidualDropPath.apply(ad, yd, 0.25, key)
    out.backward(torch.ones_like(out))
    assert relerr(out.detach().cpu(), a + y * keep / 0.75) <= 1e-6
    assert torch.equal(ad.grad.cpu(), torch.ones(6, 10, 64)) and relerr(yd.grad.cpu(), (keep / 0.75).expand(6, 10, 64)) <= 1e-6


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("B,N,Ci,Co,drop_p,dp_p", [(3, 200, 64, 320, 0.3, 0.25), (2, 131, 1280, 320, 0.3, 0.1), (4, 128, 320, 320, 0.0, 0.0),
                                                    (2, 400, 128, 128, 0.0, 0.4), (5, 160, 36, 32, 0.3, 0.0)])
def test_linear_with_the_residual_in_its_epilogue(B, N, Ci, Co, drop_p, dp_p, mode):
    """Round 5: out = res + DropPath(Dropout(linear(x))) with the add in the 1x1 GEMM kernel's epilogue (fs_conv2d_fwd_residual) and one mask
    pass in the backward (fs_droppath_dropout_bwd), against the composition it replaces (ConvBias with its fused Dropout, then
    ResidualDropPath): same masks -- the zeros coincide -- same values to rounding, same gradients; 131 tokens per sample puts sample
    boundaries inside the 128-row tiles."""
    fovealseg.hip.set_conv_precision(mode)
    try:
        g = torch.Generator().manual_seed(B * 100 + N + Ci)
        x = torch.randn(B, N, Ci, generator=g).to(DEV)
        res = torch.randn(B, N, Co, generator=g).to(DEV)
        w = rsck_param(torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5)
        b = (torch.randn(Co, generator=g) * 0.1).to(DEV)
        cot = torch.randn(B, N, Co, generator=g).to(DEV)
        k1, k2 = ops.layer_key(7, 11), ops.layer_key(7, 12)

        def run(fused):
            xs, rs, ws_, bs = x.clone().requires_grad_(True), res.clone().requires_grad_(True), w.detach().clone().requires_grad_(True), b.clone().requires_grad_(True)
            ws_ = rsck_param(w.detach().cpu()).requires_grad_(True)
            if fused:
                out = ops.linear_residual(xs, ws_, bs, rs, drop_p, k1 if drop_p > 0 else 0, dp_p, k2 if dp_p > 0 else 0)
                assert out is not None
            else:
                y = ops.ConvBias.apply(xs.reshape(1, -1, 1, Ci), ws_, bs, 1, 0, drop_p, k1 if drop_p > 0 else 0).view(B, N, Co)
                out = ops.ResidualDropPath.apply(rs, y, dp_p, k2 if dp_p > 0 else 0)
            out.backward(cot)
            return out.detach(), xs.grad, rs.grad, ws_.grad, bs.grad
        fo, fx, fr, fw, fb = run(True)
        uo, ux, ur, uw, ub = run(False)
        branch_f, branch_u = fo - res, uo - res
        assert torch.equal(branch_f.abs() < 1e-12, branch_u.abs() < 1e-12) or float(((branch_f == 0) != (branch_u == 0)).float().mean()) < 1e-6
        assert float((fo - uo).abs().max()) <= 2e-6 * float(uo.abs().max())
        assert torch.equal(fr, ur) and torch.equal(fr, cot)
        assert relerr(fx, ux) <= 2e-6 and relerr(fw, uw) <= 2e-5 and relerr(fb, ub) <= 2e-5
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


# Nk = 100 / 400: the sequence-reduced key counts at 80x80 / 160x160 inputs (BASELINE configs[3]); ragged N and Nk; N < one wave tile
@pytest.mark.parametrize("heads,N,Nk,p", [(1, 200, 100, 0.0), (5, 77, 100, 0.2), (2, 130, 25, 0.2), (2, 1600, 400, 0.2), (1, 300, 7, 0.0),
                                         (8, 25, 25, 0.0), (1, 6400, 100, 0.2)])
@pytest.mark.parametrize("mode", ["f32", "bf16x3"])       # f32: the exact-fp32 MFMA kernels; bf16x3: csrc/attention_split.hip (forward)
def test_attention_vs_torch(heads, N, Nk, p, mode):
    fovealseg.hip.set_conv_precision(mode)
    try:
        _attention_vs_torch(heads, N, Nk, p)
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


def _attention_vs_torch(heads, N, Nk, p):
    g = torch.Generator().manual_seed(43)
    B, C = 2, heads * 64
    q, k, v = (torch.randn(B, n, C, generator=g) for n in (N, Nk, Nk))
    key = ops.layer_key(7, 70)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    qh, kh, vh = (t.view(B, -1, heads, 64).transpose(1, 2) for t in (qr, kr, vr))
    probs = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1)
    if p > 0:
        keep = torch.from_numpy(O.dropout_keep_mask_nhwc(B * heads * N * Nk, key, p)).view(B, heads, N, Nk).float()
        probs = probs * keep * np.float32(1.0 / (1.0 - p))
    ref = (probs @ vh).transpose(1, 2).reshape(B, N, C)
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    out = ops.Attention.apply(qd, kd, vd, heads, p, key)
    out.backward(cot.to(DEV))
    assert relerr(out.detach().cpu(), ref.detach()) <= 1e-5
    assert relerr(qd.grad.cpu(), qr.grad) <= 1e-4
    assert relerr(kd.grad.cpu(), kr.grad) <= 1e-4
    assert relerr(vd.grad.cpu(), vr.grad) <= 1e-4


def test_attention_keep_words_match_the_hash():
    """The split forward leaves one keep bit per (query, key); the split backward must give the same gradients from those words as from
    hashing the element indices again (mask = NULL), and the words must equal the oracle's replay mask."""
    H = fovealseg.hip
    H.set_conv_precision("bf16x3")
    try:
        B, heads, N, Nk, p = 2, 2, 203, 77, 0.2
        C = heads * 64
        g = torch.Generator().manual_seed(5)
        q, k, v, go = (torch.randn(B, n, C, generator=g).to(DEV) for n in (N, Nk, Nk, N))
        key = ops.layer_key(3, 30)
        o, lse = torch.empty_like(q), torch.empty(B * heads * N, device=DEV)
        nw = int(H.load().fs_attention_mask_words(B, N, Nk, heads))
        assert nw == B * heads * N * 3
        mask = torch.zeros(nw, device=DEV, dtype=torch.int32)
        nb = H.attention_split_ws_bytes(B, Nk, heads)
        ws = torch.empty(nb, device=DEV, dtype=torch.uint8)
        H.call("fs_attention_fwd_split", H.ptr(q), H.ptr(k), H.ptr(v), H.ptr(o), H.ptr(lse), H.ptr(mask), H.ptr(ws), nb, B, N, Nk, heads, 0.125, p, key)
        keep = torch.from_numpy(O.dropout_keep_mask_nhwc(B * heads * N * Nk, key, p)).view(B * heads * N, Nk)
        words = mask.view(B * heads * N, 3).cpu().numpy().astype(np.uint32)
        bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(B * heads * N, 96)[:, :Nk]
        assert np.array_equal(bits.astype(bool), keep.numpy().astype(bool))
        nbb = H.attention_split_ws_bytes(B, Nk, heads, backward=True)
        wsb = torch.empty(nbb, device=DEV, dtype=torch.uint8)
        outs = []
        for m in (mask, None):
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            scratch = torch.empty(B * heads * N, device=DEV)
            H.call("fs_attention_bwd_split", H.ptr(q), H.ptr(k), H.ptr(v), H.ptr(o), H.ptr(go), H.ptr(lse), H.ptr(m), H.ptr(dq), H.ptr(dk),
                   H.ptr(dv), H.ptr(scratch), H.ptr(wsb), nbb, B, N, Nk, heads, 0.125, p, key)
            outs.append((dq, dk, dv))
        assert torch.equal(outs[0][0], outs[1][0])                       # dq: no atomics, the same arithmetic
        for a, b_ in zip(outs[0][1:], outs[1][1:]):                      # dk / dv: split-q partial sums meet in atomics
            assert relerr(a, b_) <= 1e-6
    finally:
        H.set_conv_precision(H.default_conv_precision())


def test_attention_layer_leaves_keep_words_in_training():
    """inside Function.forward grad mode is always off, so the decision "a backward will follow" must come from the call site.
    Through the SegFormer layer in train mode the forward must leave the keep words (ctx.keep_mask) and the gradients must equal the ones
    the hash path gives (mask dropped before the backward); under no_grad / in eval mode no words are allocated."""
    from fovealseg import segformer as SF
    H = fovealseg.hip
    H.set_conv_precision("bf16x3")
    try:
        torch.manual_seed(0)
        layer = SF.EfficientSelfAttention(128, 2, 4).to(DEV)
        layer._path = "encoder.block.1.0.attention.self"
        layer.train()
        x = torch.randn(2, 20, 20, 128, device=DEV)
        grads = []
        for drop_mask in (False, True):
            xd = x.clone().requires_grad_(True)
            layer.zero_grad()
            out = layer(xd)
            node = out.grad_fn
            while node is not None and type(node).__name__ != "AttentionBackward":
                node = node.next_functions[0][0]
            assert node is not None, "the layer's graph holds no Attention node"
            assert node.keep_mask is not None and node.keep_mask.dtype == torch.int32 and node.split
            if drop_mask:
                node.keep_mask = None                       # the backward kernels hash every (query, key) element again
            out.backward(torch.ones_like(out))
            grads.append([xd.grad.clone()] + [p_.grad.clone() for p_ in layer.parameters()])
        assert torch.equal(grads[0][0], grads[1][0]) or relerr(grads[0][0], grads[1][0]) <= 1e-6
        scale = max(float(b_.abs().max()) for b_ in grads[1][1:])
        for a, b_ in zip(grads[0][1:], grads[1][1:]):        # (the key bias gradient is analytically zero -- softmax shift invariance: absolute bound)
            assert float((a - b_).abs().max()) <= 1e-5 * max(float(b_.abs().max()), 1e-3 * scale)
        with torch.no_grad():
            q, k, v = (torch.randn(2, n, 128, device=DEV).requires_grad_(True) for n in (50, 25, 25))
        # requires_grad inputs, but grad mode off at the call site: no backward will follow, no words
        class Spy:
            words = 0
        real = H.call

        def spy(name, *args):
            if name == "fs_attention_fwd_split" and args[5] is not None:
                Spy.words += 1
            return real(name, *args)
        ops.hip.call = spy
        try:
            with torch.no_grad():
                ops.attention(q, k, v, 2, 0.2, 5)
            assert Spy.words == 0
            ops.attention(q, k, v, 2, 0.2, 5)
            assert Spy.words == 1
        finally:
            ops.hip.call = real
    finally:
        H.set_conv_precision(H.default_conv_precision())


@pytest.mark.parametrize("B,H,W,Ci,Co,k,stride,pad", [(2, 23, 17, 3, 64, 7, 1, 3), (2, 24, 16, 64, 64, 8, 8, 0), (1, 29, 31, 3, 32, 7, 4, 3),
                                                      (2, 16, 16, 32, 48, 8, 8, 0),
                                                      # stride = filter size: the permutation kernels (round 5); 22 x 18 under 4 x 4 patches leaves
                                                      # two rows and two columns outside every patch (zero gradient there)
                                                      (2, 22, 18, 64, 64, 4, 4, 0), (1, 12, 10, 128, 64, 2, 2, 0), (3, 9, 11, 320, 320, 2, 2, 0)])
def test_big_filter_conv_as_unfolded_linear(B, H, W, Ci, Co, k, stride, pad):
    """Filters of more than 32 taps (SegFormer's 7x7 patch embedding, 8x8 stride-8 sequence reduction) run as unfold + one linear layer in
    bf16x3 (ops.conv_bias_any): forward, input gradient (fold), weight and bias gradients against torch in fp64."""
    fovealseg.hip.set_conv_precision("bf16x3")
    try:
        g = torch.Generator().manual_seed(B + H + Ci + k)
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
        b = torch.randn(Co, generator=g)
        x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
        ref = F.conv2d(x64, w64, b64, stride, pad)
        cot = torch.randn(ref.shape, generator=g)
        ref.backward(cot.double())
        xd = nhwc(x).requires_grad_(True)
        wd = rsck_param(w).requires_grad_(True)
        bd = b.to(DEV).requires_grad_(True)
        calls = []
        real = fovealseg.hip.call
        fovealseg.hip.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        try:
            y = ops.conv_bias_any(xd, wd, bd, stride, pad)
            y.backward(nhwc(cot))
        finally:
            fovealseg.hip.call = real
        assert "fs_unfold" in calls and "fs_fold" in calls and "fs_linear_bwd_weight_bias" in calls
        assert relerr(nchw(y.detach()), ref.detach()) <= 1e-5
        assert relerr(nchw(xd.grad), x64.grad) <= 1e-5
        assert relerr(wd.grad.cpu(), w64.grad) <= 2e-5
        assert relerr(bd.grad.cpu(), b64.grad) <= 1e-5
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


def _segformer_pair():
    import segformer_oracle as SO
    from fovealseg import segformer as S
    o = SO.OracleSegformer()
    fovealseg.weights.apply_name_keyed_init(o)
    m = S.segformer()
    m.load_state_dict(o.state_dict(), strict=True)
    return o, m.to(DEV), SO


def test_segformer_eval_vs_golden_and_oracle(golden, prec):
    g = golden("g13_segformer")
    o, m, SO = _segformer_pair()
    o.eval()
    m.eval()
    x = T(g["x"])
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd)[0]
    assert out.shape == (1, 1024, 80, 80)
    oc = out.detach().cpu()
    assert np.abs(oc[0, :, 32:48, 32:48].numpy() - g["crop"]).max() <= 2e-4          # transformers 5.15.0 golden
    assert np.abs(oc.mean(dim=(0, 2, 3)).numpy() - g["chan_mean"]).max() <= 2e-4
    xr = x.clone().requires_grad_(True)
    ref = o(xr)[0]
    cot = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)) * 0.01
    o.zero_grad()
    ref.backward(cot)
    m.zero_grad()
    out.backward(cot.to(DEV))
    assert relerr(oc, ref.detach()) <= 1e-4
    assert relerr(xd.grad.cpu(), xr.grad) <= 2e-3
    po, pm = dict(o.named_parameters()), dict(m.named_parameters())
    for k in ("segformer.encoder.patch_embeddings.0.proj.weight", "segformer.encoder.block.0.1.attention.self.sr.weight",
              "segformer.encoder.block.2.17.attention.self.key.weight", "segformer.encoder.block.1.3.mlp.dwconv.dwconv.weight",
              "segformer.encoder.block.3.2.mlp.dense2.bias", "segformer.encoder.block.2.39.layer_norm_2.weight",
              "segformer.encoder.layer_norm.1.bias"):
        assert relerr(pm[k].grad.cpu(), po[k].grad) <= 2e-3, k


def test_config3_segformer_160_vs_oracle():
    """BASELINE configs[3] size: the SegFormer encoder at task_input_size 160x160 (25 600 / 6 400 / 1 600 / 400 tokens,
    400 reduced keys per stage-1..3 layer), B = 1, eval mode, forward AND backward against oracle/segformer_oracle.py -- the restatement
    pinned to transformers 5.15.0 at 80x80 by g13_segformer.npz; w.r.t. the reference's transformers 4.46.2 it stays PARITY UNPINNED
    (package absent, the reference holds no fixture).  Library default mode (bf16x3)."""
    o, m, SO = _segformer_pair()
    o.eval()
    m.eval()
    g = torch.Generator().manual_seed(160)
    x = torch.rand(1, 3, 160, 160, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd)[0]
    assert out.shape == (1, 1024, 160, 160)
    xr = x.clone().requires_grad_(True)
    ref = o(xr)[0]
    cot = torch.randn(ref.shape, generator=g) * 0.01
    o.zero_grad()
    ref.backward(cot)
    m.zero_grad()
    out.backward(cot.to(DEV))
    assert relerr(out.detach().cpu(), ref.detach()) <= 1e-4
    assert relerr(xd.grad.cpu(), xr.grad) <= 2e-3
    po, pm = dict(o.named_parameters()), dict(m.named_parameters())
    worst = ("", 0.0)
    for k, q in pm.items():
        if q.grad is None or po[k].grad is None or k.endswith("self.key.bias"):      # key bias: analytically zero gradient
            continue
        e = relerr(q.grad.cpu(), po[k].grad)
        if e > worst[1]:
            worst = (k, e)
    print("configs[3] SegFormer 160x160 vs oracle: worst parameter-gradient max-norm error", worst)
    assert worst[1] <= 3e-3, worst


def test_segformer_layer_train_replay(prec):
    """One stage-2 transformer block in train mode: hidden/attention dropout and DropPath replayed from the hash."""
    o, m, SO = _segformer_pair()
    o.train()
    m.train()
    ops.DropoutState.seed, ops.DropoutState.step = 8, 1
    i, j = 1, 4
    ob, mb = o.segformer.encoder.block[i][j], m.segformer.encoder.block[i][j]
    base = f"segformer.encoder.block.{i}.{j}"

    def key_of(path):
        return ops.DropoutState.key(ops.layer_id_from_name(path))

    class H(SO.Hooks):
        def dropout(self, path, x, p, training):
            # module paths: <base>.attention.output.dropout, <base>.mlp.dropout1/2 (elements in (B,N,C) order)
            keep = torch.from_numpy(O.dropout_keep_mask_nhwc(x.numel(), key_of(path), p)).view(x.shape).float()
            return x * keep * np.float32(1.0 / (1.0 - p))

        def attn_dropout(self, path, probs, p, training):
            keep = torch.from_numpy(O.dropout_keep_mask_nhwc(probs.numel(), key_of(path), p)).view(probs.shape).float()
            return probs * keep * np.float32(1.0 / (1.0 - p))

        def drop_path(self, path, y, p, training):
            keep = torch.from_numpy(O.dropout_keep_mask_nhwc(y.shape[0], key_of(path), p)).float().view(-1, 1, 1)
            return y * keep * np.float32(1.0 / (1.0 - p))
    g = torch.Generator().manual_seed(47)
    B, h, w, C = 4, 12, 12, 128
    x = torch.randn(B, h * w, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = ob(xr, h, w, H(), base, True)
    cot = torch.randn(ref.shape, generator=g)
    ob.zero_grad()
    ref.backward(cot)
    xd = x.view(B, h, w, C).to(DEV).requires_grad_(True)
    out = mb(xd)
    mb.zero_grad()
    out.backward(cot.view(B, h, w, C).to(DEV))
    assert relerr(out.detach().cpu().view(B, h * w, C), ref.detach()) <= 2e-5
    assert relerr(xd.grad.cpu().view(B, h * w, C), xr.grad) <= 2e-4
    po, pm = dict(ob.named_parameters()), dict(mb.named_parameters())
    for k in po:
        if k.endswith("self.key.bias"):      # analytically zero: a constant added to every key leaves the softmax unchanged
            continue
        assert relerr(pm[k].grad.cpu(), po[k].grad) <= 5e-4, k


def test_segformer_through_module_surface():
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
    from fovealseg import train
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    batch = train.synthetic_batch(2, 128, 128, seed=9, device=DEV)
    feed = {"img_data": batch[0], "seg_label": batch[2], "focus_point": batch[1], "cls_label": batch[3]}
    loss, acc, edge = module(feed)
    loss.mean().backward()
    assert np.isfinite(float(loss.detach()))
    assert module.encoder.segformer.encoder.block[2][20].mlp.dense1.weight.grad is not None


def test_linear_weight_gradients_land_in_the_arena_through_their_views():
    """Round 5: a linear layer hands the conv engine a VIEW of its parameter (the (out, in) matrix as a 1x1 filter; a k x k stride-k filter as
    the matrix of its patch rows), so the arena-direct path did not see a leaf and every such gradient travelled ViewBackward ->
    AccumulateGrad -> add_ (370 launches per configs[3] step).  ops.param_view records how the view was made; the kernels then add into the
    same view of the arena slice.  Same gradients on both routes, and no AccumulateGrad add left for the weights of the linears."""
    from fovealseg import train
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    opts = train.create_optimizers(nets, cfg)
    batch = train.synthetic_batch(2, 128, 128, seed=9, device=DEV)

    def run(direct):
        saved = ops.DIRECT_GRAD
        ops.DIRECT_GRAD = direct
        feed = {"img_data": batch[0], "seg_label": batch[2], "focus_point": batch[1], "cls_label": batch[3]}      # (forward replaces seg_label)
        try:
            for o in opts:
                o.zero_grad()
            ops.DropoutState.seed, ops.DropoutState.step = 5, 0
            ops.reset_step_state()
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
                loss, _, _ = module(feed)
                loss.mean().backward()
                torch.cuda.synchronize()

            def is_matrix(sh):          # a linear's (out, in) weight or a k x k stride-k reduction filter (C, C, k, k); activations are (B, ..., C) with B = 2
                sh = tuple(sh)
                return (len(sh) == 2 and min(sh) >= 64) or (len(sh) == 4 and sh[0] == sh[1] >= 64 and sh[2] == sh[3] and sh[2] in (2, 4, 8))
            adds = sum(1 for e in prof.events() if e.name == "aten::add_" and e.input_shapes and is_matrix(e.input_shapes[0]))
            return [o.flat.grad.clone() for o in opts], adds
        finally:
            ops.DIRECT_GRAD = saved
    g_direct, adds_direct = run(True)
    g_auto, adds_auto = run(False)
    for a, b in zip(g_direct, g_auto):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-30)        # bwd-weight's float atomics
    blk = module.encoder.segformer.encoder.block
    n_lin = sum(6 for st in blk for b_ in st) + sum(1 for st in blk for b_ in st if b_.attention.self.sr_ratio > 1)
    assert adds_auto >= n_lin and adds_direct == 0, (adds_auto, adds_direct, n_lin)


# ------------------------------------------------------------------------------------------------
# BASELINE configs[3] / configs[4]: task network at 160x160 on an 80x80 saliency grid (grid up-sampling,
# models/models.py:621-631); DeepLab behind a 2048x2048 -> 80x80 warp at the per-GPU batch of the 8-GPU config.
# "parity unpinned" for the SegFormer / DeepLab encoders themselves (see their oracle headers); what is checked here
# is the grid path against torch, and that the named configurations RUN end to end (forward, backward, finite, shapes).
# ------------------------------------------------------------------------------------------------
def test_grid_upsample_vs_torch():
    g = torch.Generator().manual_seed(5)
    for (h, w, H, W) in ((80, 80, 160, 160), (20, 30, 60, 60), (16, 16, 16, 16)):
        grid = torch.rand(3, h, w, 2, generator=g) * 2 - 1
        gr = grid.clone().requires_grad_(True)
        ref = F.interpolate(gr.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        cot = torch.randn(ref.shape, generator=g)
        ref.backward(cot)
        gd = grid.to(DEV).requires_grad_(True)
        out = ops.GridUpsample.apply(gd, H, W)
        out.backward(cot.to(DEV))
        assert out.shape == (3, H, W, 2)
        assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 1e-6
        assert relerr(gd.grad.cpu(), gr.grad) <= 1e-5


def test_config3_segformer_task160():
    """BASELINE configs[3]: SegFormer encoder (fc_dim 1024), task_input_size (160,160) on the (80,80) saliency grid."""
    from fovealseg import train
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
    cfg.TRAIN.task_input_size = (160, 160)
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    X, Fp, Y, cls = train.synthetic_batch(2, 512, 512, seed=9, device=DEV)
    feed = {"img_data": X, "seg_label": Y, "focus_point": Fp, "cls_label": cls}
    xs, _ = module.saliency(X, Fp)
    grid = module.create_grid(xs)
    assert grid.shape == (2, 160, 160, 2)
    # the up-sampled grid against torch on the 80x80 grid the module computes
    g80 = ops.GaussGrid.apply(xs, module.g1d, module.padding_size_x).detach().cpu()
    ref = F.interpolate(g80.permute(0, 3, 1, 2), size=(160, 160), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    assert float((grid.detach().cpu() - ref).abs().max()) <= 1e-6
    loss, acc, edge = module(feed)
    loss.mean().backward()
    assert feed["seg_label"].shape == (2, 160, 160) and feed["seg_label"].dtype == torch.int64
    assert np.isfinite(float(loss.detach())) and 0.0 <= float(acc) <= 1.0
    for n, p_ in module.named_parameters():
        if p_.requires_grad and not n.startswith("encoder.decode_head"):
            assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n
    assert float(module.localization.fov_expand_1.weight.grad.abs().max()) > 0     # the gradient reaches the saliency net through the up-sampled grid
    # evaluation caller on the same configuration
    module.eval()
    with torch.no_grad():
        outs = module({"img_data": X, "seg_label": Y, "focus_point": Fp, "cls_label": cls}, is_inference=True)
    assert len(outs) == 6 and all(np.isfinite(float(o)) for o in outs)


@pytest.mark.parametrize("B", [16, 64])
def test_config3_segformer_at_its_real_size(B):
    """configs[3]'s stated workload through the real module -- SegFormer encoder (fc_dim 1024), 1024 x 1024 input, task size
    160 x 160, B = 16 per GPU; B = 64 takes the >= 4 GB batch-range path of the conv entry points (the C1 head's 1024-channel input at
    160 x 160 is 6.7 GB) and needs ~180 GB of HBM, so it is skipped on a smaller card.  Size-independent properties: finite loss, accuracies
    in [0, 1], every parameter gradient present and finite, the saliency net reached through the up-sampled grid, label map int64 of the
    task size, four optimiser steps that move every arena, batch-range path really taken."""
    from fovealseg import train
    free, total = torch.cuda.mem_get_info()
    if B == 64 and total < 230 * 2 ** 30:
        pytest.skip("B = 64 at 1024^2 / 160^2 needs ~180 GB")
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
    cfg.TRAIN.task_input_size = (160, 160)
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    opts = train.create_optimizers(nets, cfg)
    batch = train.synthetic_batch(B, 1024, 1024, seed=21, device=DEV)
    ranges = []
    real = ops._batch_ranges

    def spy(Bn, *per):
        r = real(Bn, *per)
        ranges.append(len(r))
        return r
    ops._batch_ranges = spy
    try:
        ops.DropoutState.seed, ops.DropoutState.step = 17, 0
        start = [op.flat.data.clone() for op in opts]
        out0 = train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=0)
        l0 = float(out0[0].detach())
        for n, p_ in module.named_parameters():
            if p_.requires_grad and not n.startswith("encoder.decode_head"):
                assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n
        assert float(module.localization.fov_expand_1.weight.grad.abs().max()) > 0
        losses = [l0]
        for it in range(1, 4):
            losses.append(float(train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=it)[0].detach()))
    finally:
        ops._batch_ranges = real
    assert all(np.isfinite(v) for v in losses), losses
    assert 0.0 <= float(out0[1]) <= 1.0 and np.isfinite(float(out0[2]))
    assert all(float((op.flat.data - w0).abs().max()) > 0 for op, w0 in zip(opts, start)), "an optimiser did not move its parameters"
    assert (max(ranges) > 1) == (B == 64), (B, max(ranges))            # tensors of 4 GB and more are convolved in batch ranges
    module.eval()
    X, Fp, Y, cls = batch
    feed = {"img_data": X[:, :3], "seg_label": Y, "focus_point": Fp, "cls_label": cls}
    with torch.no_grad():
        outs = module(feed, is_inference=True)
    assert feed["seg_label"].shape == (B, 160, 160) and feed["seg_label"].dtype == torch.int64
    assert len(outs) == 6 and all(np.isfinite(float(o)) for o in outs)
    del module, nets, opts, batch
    torch.cuda.empty_cache()


def test_config4_deeplab_2048():
    """BASELINE configs[4] at its per-GPU size: DeepLab encoder, 2048x2048 input -> (80,80) foveated warp, batch 16."""
    from fovealseg import train
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.arch_encoder = "deeplab"
    module, nets = train.build_module(cfg, device=DEV)
    module.train()
    B, H = 16, 2048
    X, Fp, Y, cls = train.synthetic_batch(B, H, H, seed=4, device=DEV)
    # front-end properties at this size (size-independent): constant image samples to the constant, label map is binary,
    # area pooling of the disc mask keeps its mean
    xs, x_low = module.saliency(X, Fp)
    assert x_low.shape == (B, 80, 80, 5) and abs(float(xs.sum()) - B) <= 1e-3
    grid = module.create_grid(xs)
    ones = ops.GridSample.apply(torch.ones(B, 3, H, H, device=DEV), grid.detach())
    inside = (grid.detach().abs() < 0.999).all(-1)
    assert float((ones[inside] - 1).abs().max()) <= 2e-7
    lab = ops.grid_sample_label(Y, grid.detach())
    assert set(torch.unique(lab).tolist()) <= {0, 1}
    ap = ops.area_pool(Y, 80, 80)
    assert abs(float(ap.mean()) - float(Y.mean())) <= 1e-5
    feed = {"img_data": X, "seg_label": Y, "focus_point": Fp, "cls_label": cls}
    loss, acc, edge = module(feed)
    loss.mean().backward()
    assert feed["seg_label"].shape == (B, 80, 80)
    assert np.isfinite(float(loss.detach())) and np.isfinite(float(edge.detach()))
    for n, p_ in module.named_parameters():
        if p_.requires_grad:
            assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n


def test_config2_hrnet_640_per_gpu_shape():
    """BASELINE configs[2] at its per-GPU shape: HRNetV2, LVIS-50 'sp60000' geometry (640x640 zero-padded frames), batch 32 per
    GPU (256 global over 8 ranks; the ranks differ only in their shard, tests/test_ddp_gloo.py covers the exchange).  One full
    optimisation step through train_step in the headline arithmetic, then the evaluation caller on the same batch."""
    from fovealseg import train
    fovealseg.hip.set_conv_precision("bf16x3")
    try:
        cfg = fovealseg.lvis50_cfg()
        module, nets = train.build_module(cfg, device=DEV)
        module.train()
        opts = train.create_optimizers(nets, cfg)
        try:
            batch = train.synthetic_batch(32, 640, 640, seed=2, device=DEV)
            p0 = opts[0].flat.data.clone()
            out = train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=0)
            torch.cuda.synchronize()
            loss, acc, edge = (float(o.detach()) for o in out)
            assert np.isfinite(loss) and np.isfinite(edge) and 0.0 <= acc <= 1.0
            assert bool(torch.isfinite(opts[0].flat.grad).all()) and float(opts[0].flat.grad.abs().max()) > 0
            assert float((opts[0].flat.data - p0).abs().max()) > 0            # the encoder arena stepped
            module.eval()
            outs = train.eval_step(module, batch)
            assert len(outs) == 6 and all(np.isfinite(float(o)) for o in outs)
        finally:
            pass          # (round 1 reset a process-global here; the direct-gradient decision is per parameter now)
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


def test_dice_known_answers_unpinned_toolbelt():
    """The hand-derived Dice known answers (A18, toolbelt absent -> parity otherwise unpinned) through the HIP loss kernel."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_dice_unpinned.json")) as f:
        kat = json.load(f)
    for c in kat["cases"]:
        pred = torch.tensor(c["logits"], dtype=torch.float32, device=DEV)
        gt = torch.tensor(c["gt"], dtype=torch.int64, device=DEV)
        out = ops.SegLoss.apply(pred, gt, 5.0)
        assert abs(float(out[2]) - c["dice"]) <= 2e-7, (c["name"], float(out[2]), c["dice"])


@pytest.mark.parametrize("M,C", [(409600, 64), (1000, 512), (77, 1024), (300, 51), (4096, 1280), (5, 4), (12345, 240)])
def test_colsum_bias_gradient(M, C):
    """fs_colsum (bias gradients): the 16-byte RowWalk kernel (C % 4 == 0, C <= 1024) and the scalar fallback, ragged row counts."""
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    got = ops.colsum(x.to(DEV), C).cpu()
    want = x.double().sum(0)
    assert float((got.double() - want).abs().max()) <= 2e-6 * max(1.0, float(x.abs().sum(0).max()))


@pytest.mark.parametrize("n", [2, 3, 4, 6])
def test_fan_out_gradient_sum_bit_exact(n):
    """ops.fan_out: the consumers' gradients are summed by fs_add_n (up to four operands per launch) -- same left-to-right fp32 order
    as the chain of binary adds the autograd engine would have issued, so bit-exact against it."""
    g = torch.Generator().manual_seed(n)
    x = torch.randn(3, 40, 24, 20, generator=g).to(DEV).requires_grad_(True)
    ws = [torch.randn(3, 40, 24, 20, generator=g).to(DEV) for _ in range(n)]
    parts = ops.fan_out(x, n)
    assert all(p.data_ptr() == x.data_ptr() for p in parts)
    sum((p * w).sum() for p, w in zip(parts, ws)).backward()
    want = ws[0].clone()
    for w in ws[1:]:
        want = want + w
    assert torch.equal(x.grad, want)


# F(2,3) row-transform kernel (csrc/conv_wino.hip): even widths, ragged tiles of the stacked batch, channel counts that are not
# multiples of the 32-k chunk / the 64-column workgroup, bias, BatchNorm partial sums, dropout replay, bwd-data.
WINO_CASES = [
    # mode, B, H, W, Cin, Cout
    ("bf16x3", 2, 7, 6, 32, 64),
    ("bf16x3", 3, 9, 4, 36, 48),        # K tail (36 = 32 + 4), N tail
    ("bf16x3", 1, 5, 10, 64, 240),      # 4 column blocks, the last one 48 wide; 5 pairs per row
    ("bf16x3", 5, 3, 8, 40, 20),
    ("bf16x3", 2, 33, 34, 64, 64),      # odd pair count per row (17): ragged tiles in x, tiles that straddle two images in y
    ("f16x2", 2, 6, 8, 128, 64),        # f16x2 takes this kernel from 128 input channels up
    ("f16x2", 3, 11, 12, 160, 96),
    ("f16x2", 1, 20, 20, 256, 128),
    # F(4,3) along the row (csrc/conv_wino4.hip, round 5): bf16x3, widths that are multiples of 4 -- every tile plan the HRNet maps
    # produce (80 / 40 / 20 wide: 2, 2 and 5 quads per tile row), ragged quad counts, K and N tails, tiles that straddle images
    ("bf16x3", 2, 80, 80, 64, 64),
    ("bf16x3", 3, 40, 40, 128, 128),
    ("bf16x3", 5, 20, 20, 96, 160),
    ("bf16x3", 2, 16, 24, 36, 100),     # 6 quads per row, K tail (36 = 32 + 4), N tail
    ("bf16x3", 1, 12, 8, 32, 32),
    ("bf16x3", 3, 7, 28, 64, 48),       # 7 quads per row (prime): ragged tiles in x
    # ... and its eight-wave form (128-column workgroups, the next chunk's split inside the MFMA phase): > 64 destination channels that
    # fill 128-column tiles, >= 192 source channels
    ("bf16x3", 5, 20, 20, 256, 256),
    ("bf16x3", 2, 40, 40, 288, 200),    # K tail (288 = 9 x 32), N tail (200 of 256)
    ("bf16x3", 1, 12, 8, 320, 72),      # one tile, 72 of 128 columns
    ("bf16x3", 2, 16, 24, 192, 240),    # forward eight-wave (192 -> 240), bwd-data four-wave (240 -> 192: 192 pads to 256 in 128-column tiles)
]


def _wino_choice(mode, W, Cs, Cd):
    """Kernel-choice code of a 3x3 stride-1 problem that takes the row-transform family: 8 = F(4,3) (bf16x3, W a multiple of 4 from 8 up;
    its four- or eight-wave form), 5 = F(2,3)."""
    if os.environ.get("FS_WINO4", "1") == "0" or mode != "bf16x3" or W % 4 or W < 8:
        return 5
    return 8


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_row_kernel(case):
    mode, B, H, W, Ci, Co = case
    fovealseg.hip.set_conv_precision(mode)
    try:
        lib = fovealseg.hip.load()
        ws = fovealseg.hip.conv_workspace_bytes(H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0)
        if os.environ.get("FS_WINOGRAD", "1") != "0":
            assert lib.fs_conv2d_kernel_choice(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 0, ws) == _wino_choice(mode, W, Ci, Co)
            wsb = fovealseg.hip.conv_workspace_bytes(H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 1)
            # bwd-data: the source is dY with Cout channels (>= 32 for the halo family, >= 128 for this kernel in f16x2)
            want = (_wino_choice(mode, W, Co, Ci) if mode == "bf16x3" or Co >= 128 else 2) if Co >= 32 else 3
            assert lib.fs_conv2d_kernel_choice(B, H, W, Ci, H, W, Co, 3, 3, 1, 1, 1, 1, wsb) == want
        g = torch.Generator().manual_seed(B * 1000 + H * 100 + W + Ci + Co)
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
        b = torch.randn(Co, generator=g)
        x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
        y64 = F.conv2d(x64, w64, b.double(), 1, 1)
        cot = torch.randn(y64.shape, generator=g)
        y64.backward(cot.double())
        xd, wd, bd, dyd = nhwc(x), rsck_param(w), b.to(DEV), nhwc(cot)
        # forward + BatchNorm partial sums: the slabs add up to the column sums / sums of squares of what was written
        y, slab, nwg = ops.conv2d_fwd_stats(xd, wd, bd, 1, 1)
        # F(4,3)'s output transform carries coefficients up to 8: measured 5e-7 .. 8e-7 forward and 1.0e-6 .. 2.2e-6 bwd-data (the cotangent
        # is not rectified) where F(2,3) gives 3e-7 .. 6.5e-7 / 4.6e-7 .. 9.4e-7 (profiles/r05/wino4_ab.txt); same bound forward, 5e-6 bwd-data
        assert relerr(nchw(y), y64.detach()) <= 3e-6
        sums = slab.view(nwg, Co, 2).double().sum(0).cpu()
        yf = y.double().reshape(-1, Co).cpu()
        assert float((sums[:, 0] - yf.sum(0)).abs().max()) <= 2e-5 * float(yf.abs().sum(0).max())
        assert float((sums[:, 1] - (yf * yf).sum(0)).abs().max()) <= 2e-5 * float((yf * yf).sum(0).max())
        assert torch.equal(ops.conv2d_fwd(xd, wd, bd, 1, 1), y)
        # dropout in the epilogue: the kept set is the integer hash of the NHWC element index (oracle restatement), survivors scaled
        p, key = 0.25, 12345
        yd = ops.conv2d_fwd(xd, wd, bd, 1, 1, drop_p=p, drop_key=key)
        keep = torch.from_numpy(O.dropout_keep_mask_nhwc(y.numel(), key, p)).view(y.shape).to(DEV)
        assert torch.equal(yd, torch.where(keep, y * (1.0 / (1.0 - p)), torch.zeros_like(y)))
        # bwd-data = the same kernel on the flipped, transposed weights
        dx = ops.conv2d_bwd_data(dyd, wd, xd.shape, 1, 1)
        assert relerr(nchw(dx), x64.grad) <= (5e-6 if _wino_choice(mode, W, Co, Ci) == 8 else 3e-6)
        # zeros in -> exact zeros out; a power-of-two scale goes through bit for bit where no tile mixes scales (bf16x3: everywhere)
        assert float(ops.conv2d_bwd_data(torch.zeros_like(dyd), wd, xd.shape, 1, 1).abs().max()) == 0.0
        if mode == "bf16x3":
            assert torch.equal(ops.conv2d_bwd_data(dyd * 2.0 ** 20, wd, xd.shape, 1, 1), dx * 2.0 ** 20)
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())


@pytest.mark.parametrize("C,hw,act_last", [(64, 20, True), (24, 9, False), (256, 6, True), (128, 40, True)])
def test_bn_backward_sums_from_the_gradient_producer(C, hw, act_last):
    """Round 3: where the gradient of a conv + BatchNorm + activation output is formed by ops.FanOut's n-ary add, that add also writes
    the layer's BatchNorm-backward column sums (fs_add_n_bnsum) and the layer's own reduction pass (fs_bn_bwd_partial) is skipped.
    Both routes must give the same gradients: two residual blocks back to back (block output -> FanOut -> next conv + residual),
    train mode, dropout on; the unfused route is the one test_conv_bn_act / G7 pin to the reference."""
    from fovealseg import modules as Mods
    torch.manual_seed(C + hw)
    blocks = [Mods.BasicBlock(C).to(DEV) for _ in range(2)]
    for i, b in enumerate(blocks):
        b._path = f"t.{i}"
        b.train()
        with torch.no_grad():
            for bn in (b.bn1, b.bn2):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_(0.0, 0.3)
    x0 = torch.randn(3, hw, hw, C, device=DEV)
    cot = torch.randn(3, hw, hw, C, device=DEV)
    ops.DropoutState.seed, ops.DropoutState.step = 11, 3

    def run(fused):
        ops.FUSE_BN_BWD_SUMS = fused
        ops.reset_step_state()
        for b in blocks:
            b.zero_grad()
        x = x0.clone().requires_grad_(True)
        h = blocks[0](x)
        xa, xb = ops.fan_out(blocks[1](h), 2)                   # the LAST block's output feeds two consumers as well
        out = (xa * cot).sum() + (xb * xb).sum() * 0.5
        calls = []
        orig = fovealseg.hip.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)
        fovealseg.hip.call = spy
        try:
            out.backward()
        finally:
            fovealseg.hip.call = orig
        grads = [x.grad.clone()] + [p.grad.clone() for b in blocks for p in b.parameters()]
        return grads, calls
    try:
        g_fused, calls_fused = run(True)
        g_plain, calls_plain = run(False)
    finally:
        ops.FUSE_BN_BWD_SUMS = True
    # 4 BatchNorm layers.  bn2 of both blocks receives its gradient from a FanOut add (fs_add_n_bnsum); bn1 of both blocks from conv2's
    # bwd-data, whose epilogue forms the sums where the F(2,3) kernel runs (fs_conv2d_bwd_data_bnsum: even width, >= 32 channels)
    ws = fovealseg.hip.conv_workspace_bytes(hw, hw, C, hw, hw, C, 3, 3, 1, 1, 1, 1)
    conv_fuses = fovealseg.hip.bwd_data_bnsum_slabs(3, hw, hw, C, hw, hw, C, 3, 3, 1, 1, 1, ws) > 0
    assert conv_fuses == (C >= 32 and hw % 2 == 0)
    assert calls_plain.count("fs_bn_bwd_partial") == 4 and calls_plain.count("fs_add_n_bnsum") == 0 and calls_plain.count("fs_conv2d_bwd_data_bnsum") == 0
    # ... and where it does, conv1's bwd-data epilogue also ABSORBS the residual branch's gradient (the other alias of the block input's
    # fan-out): no n-ary add at the block inputs, no materialised dres; block 1's conv1 then forms block 0's bn2 sums as well
    assert calls_fused.count("fs_add_n_bnsum") == (1 if conv_fuses else 2)          # the last block's output still has two plain consumers
    assert calls_fused.count("fs_add_n") == 0 if conv_fuses else True
    assert calls_fused.count("fs_conv2d_bwd_data_bnsum") == (4 if conv_fuses else 0)
    assert calls_fused.count("fs_bn_bwd_partial") == (0 if conv_fuses else 2)
    assert not ops.BN_SLABS and not ops.PENDING_RES                 # every slab and every stashed residual gradient was consumed
    for a, b in zip(g_fused, g_plain):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale          # same sums, different summation order (and bwd-weight atomics)


BNSUM_EPILOGUE_CASES = [
    # kind, B, H, W, Cin, Cout, mask, addend
    ("1x1", 2, 20, 20, 64, 256, True, False),        # HRNet layer1 conv3's bwd-data -> bn2 (64-column workgroups)
    ("1x1", 3, 9, 7, 128, 64, True, True),           # 128-column workgroups, ragged last row tile, residual addend
    ("1x1", 1, 5, 5, 36, 32, False, False),          # ragged columns, BatchNorm without activation bits
    ("s2", 2, 20, 20, 64, 64, True, False),          # fuse down-path 0 -> 2, second convolution
    ("s2", 3, 13, 9, 128, 256, True, True),          # odd sizes: the last dX row / column has no parity-1 partner
    ("s2", 1, 6, 6, 36, 16, False, False),
]


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("case", BNSUM_EPILOGUE_CASES)
def test_bwd_data_epilogue_sums_on_the_pointwise_and_stride2_kernels(case, mode):
    """Round 5: the 1x1 GEMM kernel and the one-launch 3x3 / stride-2 bwd-data kernel form the BatchNorm-backward column
    sums of the layer that produced x in their epilogue (and take a second gradient there), like the 3x3 stride-1 kernels since round 3.
    Against the same entry point without the extras: dX (+ the masked addend) bit for bit, the slab rows add up to sum dz and
    sum dz * zhat with dz = dX masked by the activation bits."""
    kind, B, H, W, Cin, Cout, with_mask, with_add = case
    fovealseg.hip.set_conv_precision(mode)
    try:
        g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
        R, stride, pad = (1, 1, 0) if kind == "1x1" else (3, 2, 1)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        w = rsck_param(torch.randn(Cout, Cin, R, R, generator=g) / (R * Cin ** 0.5))
        dy = torch.randn(B, Ho, Wo, Cout, generator=g).to(DEV)
        y = torch.randn(B, H, W, Cin, generator=g).to(DEV)
        mean = torch.randn(Cin, generator=g).to(DEV) * 0.1
        invstd = (torch.rand(Cin, generator=g) + 0.5).to(DEV)
        bits = (torch.rand(B, H, W, Cin, generator=g) < 0.6).to(DEV)
        weights = (1 << torch.arange(4, device=DEV, dtype=torch.int32))
        amask = (bits.view(-1, 4).to(torch.int32) * weights).sum(1).to(torch.uint8) if with_mask else None
        add = torch.randn(B, H, W, Cin, generator=g).to(DEV) if with_add else None
        abits = (torch.rand(B, H, W, Cin, generator=g) < 0.5).to(DEV)
        add_mask = (abits.view(-1, 4).to(torch.int32) * weights).sum(1).to(torch.uint8) if with_add else None
        ws = fovealseg.hip.conv_workspace_bytes(H, W, Cin, Ho, Wo, Cout, R, R, stride, pad, 1, 1)
        rows = fovealseg.hip.bwd_data_bnsum_slabs(B, H, W, Cin, Ho, Wo, Cout, R, R, stride, pad, 1, ws)
        assert rows > 0
        plain = ops.conv2d_bwd_data(dy, w, (B, H, W, Cin), stride, pad)
        ops.BN_SLABS.clear()
        calls = []
        orig = fovealseg.hip.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)
        fovealseg.hip.call = spy
        try:
            dx = ops.conv2d_bwd_data(dy, w, (B, H, W, Cin), stride, pad, src_bn=(y, amask, mean, invstd, 1 if with_mask else 0),
                                     addend=(add, add_mask) if with_add else None)
        finally:
            fovealseg.hip.call = orig
        assert calls.count("fs_conv2d_bwd_data_bnsum") == 1 and calls.count("fs_conv2d_bwd_data") == 0
        want = plain + (add * abits if with_add else 0.0)
        assert torch.equal(dx, want) if not with_add else float((dx - want).abs().max()) <= 1e-6 * float(want.abs().max())
        slab, nrows, owner = ops.BN_SLABS.pop((dx.data_ptr(), y.data_ptr()))
        assert nrows == rows and owner is dx and not ops.BN_SLABS
        sums = slab.view(rows, Cin, 2).double().sum(0)
        dz = (dx * bits if with_mask else dx).double().reshape(-1, Cin)
        zhat = ((y - mean) * invstd).double().reshape(-1, Cin)
        s1, s2 = dz.sum(0), (dz * zhat).sum(0)
        assert float((sums[:, 0] - s1).abs().max()) <= 2e-5 * float(dz.abs().sum(0).max())
        assert float((sums[:, 1] - s2).abs().max()) <= 2e-5 * float((dz * zhat).abs().sum(0).max())
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())
        ops.BN_SLABS.clear()


@pytest.mark.parametrize("kind,hw", [("bottleneck", 12), ("down_path", 16), ("down_path", 13)])
def test_bottleneck_and_down_path_sums_come_from_the_bwd_data_kernels(kind, hw):
    """Round 5: the layers whose output gradient leaves a 1x1 or a 3x3 / stride-2 bwd-data kernel (HRNet layer1's conv2 -> conv3, the first
    convolution of a two-step fuse down-path) no longer run their own BatchNorm reduction pass.  Same gradients either way."""
    from fovealseg import modules as Mods
    torch.manual_seed(hw)
    if kind == "bottleneck":
        net = Mods.Bottleneck(64, 32, True).to(DEV)
        x0 = torch.randn(2, hw, hw, 64, device=DEV)
        fused_partials, plain_partials = 2, 4          # bn3 and the shortcut's BatchNorm get their gradient from the loss / bn3's apply pass
    else:
        net = Mods._Chain(Mods._ConvBn(32, 32, 3, 2, True), Mods._ConvBn(32, 64, 3, 2, False)).to(DEV)
        x0 = torch.randn(2, hw, hw, 32, device=DEV)
        fused_partials, plain_partials = 1, 2
    net.train()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, Mods.HipBatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.3)
    cot = None

    def run(fused):
        nonlocal cot
        ops.FUSE_BN_BWD_SUMS = fused
        ops.reset_step_state()
        net.zero_grad()
        x = x0.clone().requires_grad_(True)
        out = net(x)
        if cot is None:
            cot = torch.randn_like(out)
        calls = []
        orig = fovealseg.hip.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)
        fovealseg.hip.call = spy
        try:
            (out * cot).sum().backward()
        finally:
            fovealseg.hip.call = orig
        return [x.grad.clone()] + [p.grad.clone() for p in net.parameters()], calls
    try:
        g_fused, calls_fused = run(True)
        g_plain, calls_plain = run(False)
    finally:
        ops.FUSE_BN_BWD_SUMS = True
    assert calls_plain.count("fs_bn_bwd_partial") == plain_partials and calls_plain.count("fs_conv2d_bwd_data_bnsum") == 0
    assert calls_fused.count("fs_bn_bwd_partial") == fused_partials
    assert calls_fused.count("fs_conv2d_bwd_data_bnsum") == plain_partials - fused_partials
    assert not ops.BN_SLABS and not ops.PENDING_RES
    for a, b in zip(g_fused, g_plain):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-30)


@pytest.mark.parametrize("chans,hw", [((32, 64), 16), ((32, 64, 128), 16), ((16, 32, 64, 128), 24)])
def test_fuse_row_gradients_carry_the_batchnorm_sums_of_their_layers(chans, hw):
    """Round 5: the last ConvBn of every HRNet fuse path has no activation, so the fuse node's backward produces its output
    gradient directly -- the ReLU-masked fuse gradient for the down-paths (ONE tensor shared by up to three layers), its up-sampling adjoint
    for the 1x1 up-paths -- and now forms those layers' BatchNorm-backward column sums in the same launches (fs_relu_bwd_bnsum,
    fs_upsample_slice_bwd_bnsum).  Same gradients as with every layer on its own reduction pass; none of those passes left for the fuse layers."""
    from fovealseg import modules as Mods
    torch.manual_seed(sum(chans) + hw)
    saved = Mods.PARALLEL_BRANCHES
    Mods.PARALLEL_BRANCHES = False
    mod = Mods.HighResolutionModule(list(chans)).to(DEV)
    Mods._assign_paths(mod, "t")
    mod.train()
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, Mods.HipBatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.3)
    n = len(chans)
    xs0 = [torch.randn(2, hw >> i, hw >> i, c, device=DEV) for i, c in enumerate(chans)]
    cots = [torch.randn(2, hw >> i, hw >> i, c, device=DEV) for i, c in enumerate(chans)]
    ops.DropoutState.seed, ops.DropoutState.step = 5, 2

    def run(fused):
        ops.FUSE_BN_BWD_SUMS = fused
        ops.reset_step_state()
        mod.zero_grad()
        xs = [x.clone().requires_grad_(True) for x in xs0]
        outs = mod(xs)
        loss = sum((o * c).sum() for o, c in zip(outs, cots))
        calls = []
        orig = fovealseg.hip.call

        def spy(name, *a):
            calls.append(name)
            return orig(name, *a)
        fovealseg.hip.call = spy
        try:
            loss.backward()
        finally:
            fovealseg.hip.call = orig
        return [x.grad.clone() for x in xs] + [p.grad.clone() for p in mod.parameters()], calls
    try:
        g_fused, calls_fused = run(True)
        g_plain, calls_plain = run(False)
    finally:
        ops.FUSE_BN_BWD_SUMS = True
        Mods.PARALLEL_BRANCHES = saved
    downs = n * (n - 1) // 2                       # fuse terms j < i: one activation-free last ConvBn each
    ups = n * (n - 1) // 2                         # fuse terms j > i: one 1x1 ConvBn each
    assert calls_fused.count("fs_upsample_slice_bwd_bnsum") == ups and calls_fused.count("fs_upsample_slice_bwd") == 0
    assert calls_fused.count("fs_relu_bwd_bnsum") == n - 1 and calls_fused.count("fs_relu_bwd") == 1       # row 0 has no down-path
    assert calls_plain.count("fs_relu_bwd") == n and calls_plain.count("fs_upsample_slice_bwd") == ups
    # the fuse layers' own reduction passes are gone (the plain run has one per BatchNorm layer of the module)
    assert calls_plain.count("fs_bn_bwd_partial") - calls_fused.count("fs_bn_bwd_partial") >= downs + ups
    assert not ops.BN_SLABS and not ops.PENDING_RES
    for a, b in zip(g_fused, g_plain):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale


def test_branch_and_row_streams_do_not_change_the_encoder():
    """Round 5: the fuse rows of a HighResolutionModule run on the branch streams, and the modules of a stage are chained stream by stream
    (one join per stage instead of two per module).  Stream placement must not change a bit: the whole HRNetV2 encoder forward and, in
    deterministic mode, every parameter gradient are identical with the streams off, with the rows on the branch streams and with the
    stage-level chaining on top -- run twice each, so that a missing dependency (a race) would have two chances to show."""
    from fovealseg import modules as Mods
    H = fovealseg.hip
    torch.manual_seed(21)
    enc = fovealseg.ModelBuilder.build_encoder("hrnetv2_nodownsp", 960, "").to(DEV)
    enc.train()
    x0 = torch.randn(2, 64, 64, 3, device=DEV)
    saved = (Mods.PARALLEL_BRANCHES, Mods.PARALLEL_FUSE, Mods.STREAM_DEPS)
    H.set_deterministic(True)

    def run(branches, fuse, deps):
        Mods.PARALLEL_BRANCHES, Mods.PARALLEL_FUSE, Mods.STREAM_DEPS = branches, fuse, deps
        ops.reset_step_state()
        ops.DropoutState.seed, ops.DropoutState.step = 4, 9
        enc.zero_grad()
        x = x0.clone().requires_grad_(True)
        feat = enc.forward_nhwc(x)
        (feat * feat).sum().backward()
        torch.cuda.synchronize()
        return feat.detach().clone(), x.grad.clone(), [p.grad.clone() for p in enc.parameters()]
    try:
        ref = run(False, False, False)
        for cfg in ((True, False, False), (True, True, False), (True, True, True), (True, True, True)):
            got = run(*cfg)
            assert torch.equal(got[0], ref[0]), cfg
            assert torch.equal(got[1], ref[1]), cfg
            for a, b in zip(got[2], ref[2]):
                assert torch.equal(a, b), cfg
    finally:
        Mods.PARALLEL_BRANCHES, Mods.PARALLEL_FUSE, Mods.STREAM_DEPS = saved
        H.set_deterministic(False)


def test_residual_consumer_independent_of_the_conv_consumer_keeps_its_gradient():
    """a two-way fan-out whose conv alias feeds an F(2,3)-eligible 3x3 conv and whose other alias is the `res` of a layer that
    does NOT depend on that conv is unordered in the backward.  The residual layer must then materialise its gradient (never stash it for an
    epilogue that may already have run): x.grad with the fused routes on == x.grad with them off, in both evaluation orders of the loss."""
    from fovealseg import modules as M
    H = fovealseg.hip
    H.set_conv_precision("bf16x3")
    try:
        torch.manual_seed(3)
        C, hw = 64, 20
        convA, bnA = M.HipConv2d(C, C, 3, 1, 1).to(DEV), M.HipBatchNorm2d(C).to(DEV)
        convB, bnB = M.HipConv2d(C, C, 3, 1, 1).to(DEV), M.HipBatchNorm2d(C).to(DEV)
        x0 = torch.randn(2, hw, hw, C, device=DEV)
        other = torch.randn(2, hw, hw, C, device=DEV)
        cot_a, cot_b = torch.randn(2, hw, hw, C, device=DEV), torch.randn(2, hw, hw, C, device=DEV)

        def run(fused, order):
            ops.FUSE_BN_BWD_SUMS = fused
            ops.reset_step_state()
            x = x0.clone().requires_grad_(True)
            y = other.clone().requires_grad_(True)
            xa, xr = ops.fan_out(x * 1.0, 2)
            a = M.conv_bn_act(xa, convA, bnA, ops.ACT_RELU)                    # the fan-out's conv consumer
            b = M.conv_bn_act(y, convB, bnB, ops.ACT_RELU, res=xr)             # residual consumer, independent of `a`
            terms = [(a * cot_a).sum(), (b * cot_b).sum()]
            loss = terms[0] + terms[1] if order == 0 else terms[1] + terms[0]
            loss.backward()
            assert not ops.PENDING_RES
            return x.grad.clone(), y.grad.clone()
        try:
            ref = run(False, 0)
            for order in (0, 1):
                got = run(True, order)
                for g_, r_ in zip(got, ref):
                    assert relerr(g_, r_) <= 2e-5
        finally:
            ops.FUSE_BN_BWD_SUMS = True
        # a stash nobody consumed is an error at the next forward, not something that is cleared silently
        ops.PENDING_RES[12345] = (x0, None)
        with pytest.raises(H.HipLibraryError):
            ops.reset_step_state()
        assert not ops.PENDING_RES
    finally:
        H.set_conv_precision(H.default_conv_precision())


@pytest.mark.parametrize("C,hw,act,use_res", [(64, 20, 1, True), (128, 12, 2, False), (96, 10, 0, False), (64, 9, 1, True)])
def test_inference_conv_with_folded_batchnorm(C, hw, act, use_res, prec):
    """Round 3: under no_grad in eval mode the BatchNorm (running statistics), the residual add and the activation run in the conv
    epilogue where the shape's kernel has the row epilogue (fs_conv2d_fwd_affine_act) -- one launch, no intermediate conv output.
    Must equal the three-launch route (conv -> bn_eval_prepare -> bn_act_fwd) that G7 / G8 pin to the reference; odd widths and the
    f32 mode fall back to that route by themselves."""
    from fovealseg import modules as Mods
    torch.manual_seed(C * hw + act)
    conv = Mods.HipConv2d(C, C, 3, 1, 1).to(DEV)
    bn = Mods.HipBatchNorm2d(C, 0.1).to(DEV).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(2, hw, hw, C, device=DEV)
    res = torch.randn(2, hw, hw, C, device=DEV) if use_res else None
    calls = []
    orig = fovealseg.hip.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    outs = {}
    try:
        fovealseg.hip.call = spy
        for fused in (True, False):
            ops.FUSE_EVAL_BN = fused
            calls.clear()
            with torch.no_grad():
                outs[fused] = Mods.conv_bn_act(x, conv, bn, act, res=res)
            outs[("calls", fused)] = list(calls)
    finally:
        fovealseg.hip.call = orig
        ops.FUSE_EVAL_BN = True
    ws = fovealseg.hip.conv_workspace_bytes(hw, hw, C, hw, hw, C, 3, 3, 1, 1, 1, 0)
    can = fovealseg.hip.fwd_affine_act_ok(2, hw, hw, C, hw, hw, C, 3, 3, 1, 1, 1, ws)
    assert can == (prec != "f32" and hw % 2 == 0 and (prec == "bf16x3" or C >= 128))
    assert ("fs_conv2d_fwd_affine_act" in outs[("calls", True)]) == can
    assert "fs_bn_act_fwd" in outs[("calls", False)] and "fs_conv2d_fwd_affine_act" not in outs[("calls", False)]
    assert relerr(outs[True].cpu(), outs[False].cpu()) <= 5e-6
    if act == 1:
        assert float(outs[True].min()) >= 0.0
