"""CPU ORACLE for the FovealSeg hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32) restatement of the reference algorithm for the path named by
BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this file; the shipped package never does.

Pinning: the reference's own tests hold no fixture for this path.  The oracle is
therefore pinned against outputs of the reference itself, produced in the build container by
`tests/golden/make_goldens.py` (which imports /root/reference read-only) and committed as
`tests/golden/*.npz`; `tests/test_oracle_vs_golden.py` checks every stage below against them.
Two pieces are restated from third-party packages that are absent from /root/reference and not
installed: `pytorch_toolbelt==0.8.0` DiceLoss('multiclass') (requirements.txt; call site
models/models.py:482,1059) -- "parity unpinned" against the real package, the goldens embed this
restatement -- and nothing else on the HRNet path.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-5

# --------------------------------------------------------------------------------------------
# activation replay (test instrument): a ReLU's derivative is discontinuous at 0, so two correct fp32
# evaluations of the same graph can disagree on the branch an activation within rounding of 0 takes,
# and that one element moves its whole receptive field of gradients by O(1).  With ACT_REPLAY set to
# {site name: bool tensor (NCHW) "the unit under test took the active branch"}, every activation site
# below multiplies by that recorded mask instead of re-deciding the branch: the backward comparison is
# then between two smooth functions (the same way Dropout masks are replayed through _Ctx.drop_fn).
# Site names are module paths relative to the root given to assign_paths(): the BatchNorm that feeds
# the activation ("...bn1", "conv1.1", "cbr.1"), or "<HRModule path>.fuse<i>" for the fuse sums.
# --------------------------------------------------------------------------------------------
ACT_REPLAY: Optional[Dict[str, torch.Tensor]] = None
ACT_REPLAY_HI: Optional[Dict[str, torch.Tensor]] = None      # ReLU6 only: "saturated at 6" masks


def assign_paths(root: nn.Module, prefix: str = "") -> None:
    for name, m in root.named_modules():
        m._opath = (prefix + "." + name).strip(".") if name else prefix


def _site(mod: nn.Module, leaf: str) -> str:
    base = getattr(mod, "_opath", "")
    return (base + "." + leaf).strip(".")


class _MaskedPass(torch.autograd.Function):
    """t where mask else 0, forward and backward; keeps only the bool mask (1 byte per element) for the backward."""

    @staticmethod
    def forward(ctx, t, mask):
        ctx.save_for_backward(mask)
        return torch.where(mask, t, torch.zeros((), dtype=t.dtype))

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, g, torch.zeros((), dtype=g.dtype)), None


def _relu(name: str, t: torch.Tensor, six: bool = False) -> torch.Tensor:
    if ACT_REPLAY is None or name not in ACT_REPLAY:
        return F.relu6(t) if six else F.relu(t)
    out = _MaskedPass.apply(t, ACT_REPLAY[name])
    if six:
        out = out + 6.0 * ACT_REPLAY_HI[name].to(t.dtype)
    return out


class ORelu(nn.Module):
    """nn.ReLU as child i of a Sequential(..., bn, relu, ...): the site is named after child i-1, the BatchNorm in front of it."""

    def forward(self, t):
        base = getattr(self, "_opath", "")
        head, _, leaf = base.rpartition(".")
        name = (head + "." + str(int(leaf) - 1)).strip(".") if leaf.isdigit() else ""
        return _relu(name, t)


# --------------------------------------------------------------------------------------------
# constants of DeformSegmentationModule.__init__
# --------------------------------------------------------------------------------------------
def make_gaussian(size: int, fwhm: float) -> np.ndarray:
    """models/models.py:140-157 -- exp(-4 ln2 ((x-x0)^2+(y-y0)^2)/fwhm^2), float64, centre size//2."""
    ax = np.arange(0, size, 1, float)
    c = size // 2
    d2 = (ax[None, :] - c) ** 2 + (ax[:, None] - c) ** 2
    return np.exp(-4 * np.log(2) * d2 / fwhm ** 2)


def gaussian_1d(size: int, fwhm: float) -> np.ndarray:
    """Separable factor of make_gaussian: G[i,j] = g[i]*g[j]."""
    ax = np.arange(0, size, 1, float)
    return np.exp(-4 * np.log(2) * (ax - size // 2) ** 2 / fwhm ** 2)


def p_basis(hs: int, ws: int, pad_x: int, pad_y: int) -> torch.Tensor:
    """models/models.py:517-522 -- P[0,i,j]=(j-pad_y)/(ws-1), P[1,i,j]=(i-pad_x)/(hs-1) (fp32)."""
    P = torch.zeros(2, hs + 2 * pad_x, ws + 2 * pad_y, dtype=torch.float32)
    for i in range(hs + 2 * pad_x):
        for j in range(ws + 2 * pad_y):
            # python-float arithmetic then a store into an fp32 tensor, exactly as the reference loop
            P[0, i, j] = 0 * (i - pad_x) / (hs - 1.0) + (1.0 - 0) * (j - pad_y) / (ws - 1.0)
            P[1, i, j] = 1 * (i - pad_x) / (hs - 1.0) + (1.0 - 1) * (j - pad_y) / (ws - 1.0)
    return P


# --------------------------------------------------------------------------------------------
# A4/A5: gaze map + low-res 5-channel input
# --------------------------------------------------------------------------------------------
def gaze_map(focus: torch.Tensor, hs: int, ws: int) -> torch.Tensor:
    """models/models.py:684-694 + DynamicFocus/utility/torch_tools.py:65-69.
    focus (B,2) = (row, col) in [0,1) -> (B,1,hs,ws) squared normalised distance."""
    max_dist = np.sqrt(hs ** 2 + ws ** 2)
    h = focus[:, 0] * (hs - 1)
    w = focus[:, 1] * (ws - 1)
    ii = torch.arange(hs, dtype=torch.int64)[:, None].repeat(1, ws)
    jj = torch.arange(ws, dtype=torch.int64)[None, :].repeat(hs, 1)
    dist = torch.sqrt((ii[None] - h[:, None, None]) ** 2 + (jj[None] - w[:, None, None]) ** 2)
    return (dist / max_dist).unsqueeze(1) ** 2


def lowres_input(x: torch.Tensor, focus: torch.Tensor, hs: int, ws: int) -> torch.Tensor:
    """models/models.py:701-705 -- bilinear (align_corners=False) low-res RGB + 2x gaze map."""
    x_low = F.interpolate(x, size=(hs, ws), mode="bilinear")
    g = gaze_map(focus, hs, ws)
    return torch.cat((x_low, g, g), dim=1)


# --------------------------------------------------------------------------------------------
# BatchNorm with the reference's state_dict layout
# --------------------------------------------------------------------------------------------
class RefSyncBN(nn.BatchNorm2d):
    """lib/nn/modules/batchnorm.py:38-61 -- under DDP `_is_parallel` is never set, so forward is
    F.batch_norm with per-rank statistics; three extra buffers live in the state_dict (:50-54)."""

    def __init__(self, c, momentum=0.001):
        super().__init__(c, eps=BN_EPS, momentum=momentum, affine=True)
        self.register_buffer("_tmp_running_mean", torch.zeros(c))
        self.register_buffer("_tmp_running_var", torch.ones(c))
        self.register_buffer("_running_iter", torch.ones(1))


# --------------------------------------------------------------------------------------------
# A6/A7: saliency CNN + compress
# --------------------------------------------------------------------------------------------
class OracleFovSim(nn.Module):
    """saliency_network.py:302-333 (fov_simple: in 5, out 24 -> width 192)."""

    def __init__(self, cin=5, cout=24):
        super().__init__()
        w = 8 * cout
        self.fov_expand_1 = nn.Conv2d(cin, w, 3, padding=1, bias=False)
        self.fov_expand_2 = nn.Conv2d(w, w, 3, padding=1, bias=False)
        self.fov_squeeze_1 = nn.Conv2d(w, cout, 3, padding=1, bias=False)
        self.norm1 = RefSyncBN(w, momentum=0.1)
        self.norm2 = RefSyncBN(w, momentum=0.1)
        self.norm3 = RefSyncBN(cout, momentum=0.1)

    def forward(self, x):
        a = _relu(_site(self, "norm1"), self.norm1(self.fov_expand_1(x)), six=True)
        b = _relu(_site(self, "norm2"), self.norm2(self.fov_expand_2(a)), six=True)
        return self.norm3(self.fov_squeeze_1(b))


class OracleCompress(nn.Module):
    """models/models.py:360-372 -- ReLU then 1x1 conv 24->1 with bias."""

    def __init__(self, cin=24):
        super().__init__()
        self.conv_last = nn.Conv2d(cin, 1, 1)

    def forward(self, x):
        return self.conv_last(F.relu(x))


# --------------------------------------------------------------------------------------------
# A13: HRNetV2, stride-1 stem
# --------------------------------------------------------------------------------------------
DropFn = Callable[[str, torch.Tensor], torch.Tensor]


class _Ctx:
    """Carries the dropout hook and a running module path through the forward."""

    def __init__(self, training: bool, drop_fn: Optional[DropFn]):
        self.training = training
        self.drop_fn = drop_fn


def _conv(cin, cout, k, s=1, bias=False):
    return nn.Conv2d(cin, cout, k, s, k // 2, bias=bias)


class OBasic(nn.Module):
    """models/hrnetv2_nodownsp.py:32-64 -- conv1->Dropout(.3)->bn1->ReLU->conv2->Dropout(.3)->bn2->+res->ReLU."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = _conv(c, c, 3)
        self.bn1 = RefSyncBN(c, 0.1)
        self.conv2 = _conv(c, c, 3)
        self.bn2 = RefSyncBN(c, 0.1)

    def forward(self, x, ctx: _Ctx, path: str):
        def drop(name, t):
            if ctx.drop_fn is not None:
                return ctx.drop_fn(path + "." + name, t)
            return F.dropout(t, 0.3, ctx.training)
        o = _relu(_site(self, "bn1"), self.bn1(drop("conv1", self.conv1(x))))
        o = self.bn2(drop("conv2", self.conv2(o)))
        return _relu(_site(self, "bn2"), o + x)


class OBottle(nn.Module):
    """models/hrnetv2_nodownsp.py:67-105 (expansion 4, no dropout)."""

    def __init__(self, cin, planes, down):
        super().__init__()
        self.conv1 = _conv(cin, planes, 1)
        self.bn1 = RefSyncBN(planes, 0.1)
        self.conv2 = _conv(planes, planes, 3)
        self.bn2 = RefSyncBN(planes, 0.1)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = RefSyncBN(planes * 4, 0.1)
        self.downsample = nn.Sequential(_conv(cin, planes * 4, 1), RefSyncBN(planes * 4, 0.1)) if down else None

    def forward(self, x):
        o = _relu(_site(self, "bn1"), self.bn1(self.conv1(x)))
        o = _relu(_site(self, "bn2"), self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        r = x if self.downsample is None else self.downsample(x)
        return _relu(_site(self, "bn3"), o + r)


class _Seq(nn.Sequential):
    pass


def _cb(cin, cout, k, s, relu):
    mods = [_conv(cin, cout, k, s), RefSyncBN(cout, 0.1)]
    if relu:
        mods.append(ORelu())
    return nn.Sequential(*mods)


class OHRModule(nn.Module):
    """models/hrnetv2_nodownsp.py:108-252 -- n parallel branches of 4 BasicBlocks + all-to-all fuse."""

    def __init__(self, chans: List[int]):
        super().__init__()
        n = len(chans)
        self.chans = chans
        self.branches = nn.ModuleList([nn.Sequential(*[OBasic(c) for _ in range(4)]) for c in chans])
        rows = []
        for i in range(n):
            row = []
            for j in range(n):
                if j > i:                                   # :189-197  1x1 conv + BN, upsampled later
                    row.append(_cb(chans[j], chans[i], 1, 1, False))
                elif j == i:
                    row.append(None)
                else:                                       # :200-220  chain of 3x3 stride-2
                    chain = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        chain.append(_cb(chans[j], chans[i] if last else chans[j], 3, 2, not last))
                    row.append(nn.Sequential(*chain))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)

    def forward(self, xs, ctx: _Ctx, path: str):
        n = len(self.chans)
        xs = list(xs)
        for i in range(n):
            t = xs[i]
            for b, blk in enumerate(self.branches[i]):
                t = blk(t, ctx, f"{path}.branches.{i}.{b}")
            xs[i] = t
        out = []
        for i in range(n):                                  # :235-251, same left-to-right sum order
            y = xs[0] if i == 0 else self.fuse_layers[i][0](xs[0])
            for j in range(1, n):
                if j == i:
                    y = y + xs[j]
                elif j > i:
                    y = y + F.interpolate(self.fuse_layers[i][j](xs[j]), size=xs[i].shape[-2:],
                                          mode="bilinear", align_corners=False)
                else:
                    y = y + self.fuse_layers[i][j](xs[j])
            out.append(_relu(_site(self, f"fuse{i}"), y))
        return out


class OracleHRNet(nn.Module):
    """models/hrnetv2_nodownsp.py:261-446 -- widths (64,128,256,512), modules 1/4/3, concat 960."""

    WIDTHS = (64, 128, 256, 512)

    def __init__(self):
        super().__init__()
        self.conv1 = _conv(3, 64, 3)
        self.bn1 = RefSyncBN(64, 0.1)
        self.conv2 = _conv(64, 64, 3)
        self.bn2 = RefSyncBN(64, 0.1)
        self.layer1 = nn.Sequential(OBottle(64, 64, True), OBottle(256, 64, False),
                                    OBottle(256, 64, False), OBottle(256, 64, False))
        W = self.WIDTHS
        self.transition1 = nn.ModuleList([_cb(256, W[0], 3, 1, True), nn.Sequential(_cb(256, W[1], 3, 2, True))])
        self.stage2 = nn.Sequential(*[OHRModule(list(W[:2])) for _ in range(1)])
        self.transition2 = nn.ModuleList([None, None, nn.Sequential(_cb(W[1], W[2], 3, 2, True))])
        self.stage3 = nn.Sequential(*[OHRModule(list(W[:3])) for _ in range(4)])
        self.transition3 = nn.ModuleList([None, None, None, nn.Sequential(_cb(W[2], W[3], 3, 2, True))])
        self.stage4 = nn.Sequential(*[OHRModule(list(W[:4])) for _ in range(3)])

    def forward(self, x, return_feature_maps=False, drop_fn: Optional[DropFn] = None):
        ctx = _Ctx(self.training, drop_fn)
        x = _relu(_site(self, "bn1"), self.bn1(self.conv1(x)))
        x = _relu(_site(self, "bn2"), self.bn2(self.conv2(x)))
        x = self.layer1(x)
        ys = [self.transition1[0](x), self.transition1[1](x)]
        for m, mod in enumerate(self.stage2):
            ys = mod(ys, ctx, f"stage2.{m}")
        ys = [ys[0], ys[1], self.transition2[2](ys[-1])]
        for m, mod in enumerate(self.stage3):
            ys = mod(ys, ctx, f"stage3.{m}")
        ys = [ys[0], ys[1], ys[2], self.transition3[3](ys[-1])]
        for m, mod in enumerate(self.stage4):
            ys = mod(ys, ctx, f"stage4.{m}")
        size = ys[0].shape[-2:]
        ups = [ys[0]] + [F.interpolate(t, size=size, mode="bilinear", align_corners=False) for t in ys[1:]]
        return [torch.cat(ups, 1)]


# --------------------------------------------------------------------------------------------
# A15: C1 head
# --------------------------------------------------------------------------------------------
class OResBlock(nn.Module):
    """models/model_utils.py:224-246 -- biased convs, plain nn.BatchNorm2d."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1), nn.BatchNorm2d(cout), ORelu())
        self.conv2 = nn.Sequential(nn.Conv2d(cout, cout, 3, 1, 1), nn.BatchNorm2d(cout))
        self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride), nn.BatchNorm2d(cout))

    def forward(self, x):
        return _relu(_site(self, "conv2.1"), self.conv2(self.conv1(x)) + self.downsample(x))


class OClsNet(nn.Module):
    """models/model_utils.py:248-275 -- stride-4 block, stride-2 block, AvgPool(10), FC."""

    def __init__(self, cin, ncls):
        super().__init__()
        self.layer2 = nn.Sequential(OResBlock(cin, 512, 4))
        self.layer3 = nn.Sequential(OResBlock(512, 512, 2))
        self.fc = nn.Linear(512, ncls)

    def forward(self, x):
        x = self.layer3(self.layer2(x))
        if tuple(x.shape[-2:]) == (10, 10):
            x = F.avg_pool2d(x, (10, 10), stride=1)          # the reference: AvgPool2d((10,10)) -> FC(512), 80x80 inputs only
        else:
            # other input sizes fail in the reference (model_utils.py:254-255); the build pools globally (DESIGN.md "Divergences")
            x = x.mean((2, 3), keepdim=True)
        return self.fc(x.view(x.size(0), -1))


class OracleC1(nn.Module):
    """models/model_utils.py:278-309."""

    def __init__(self, num_class=51, fc_dim=960):
        super().__init__()
        self.cbr = nn.Sequential(_conv(fc_dim, fc_dim // 4, 3), RefSyncBN(fc_dim // 4, 0.001), ORelu())
        self.conv_last = nn.Conv2d(fc_dim // 4, 1, 1)
        self.cls_net = OClsNet(fc_dim, num_class)

    def forward(self, conv_out, segSize=None, res=None):
        f = conv_out[-1]
        m = torch.sigmoid(self.conv_last(self.cbr(f))) - 0.5            # (B,1,h,w)
        c = self.cls_net(f)                                             # (B,K)
        pred = c[:, :, None, None].expand(-1, -1, m.shape[2], m.shape[3]).clone()
        pred[:, -1:] = c[:, -1:, None, None] * m
        return pred


# --------------------------------------------------------------------------------------------
# A10/A11: replication pad + deformation grid
# --------------------------------------------------------------------------------------------
def create_grid(xs_hm: torch.Tensor, filt: torch.Tensor, P: torch.Tensor, hs: int, ws: int) -> torch.Tensor:
    """models/models.py:594-637 (forward branch, task size == saliency size so both Upsamples are
    identities).  xs_hm (B,1,hs+2p,ws+2p), filt (1,1,2p+1,2p+1) -> grid (B,hs,ws,2) = (x=col, y=row)."""
    B = xs_hm.shape[0]
    p = F.conv2d(xs_hm, filt)
    x_mul = (P[None] * torch.cat((xs_hm, xs_hm), 1)).view(-1, 1, xs_hm.shape[2], xs_hm.shape[3])
    allf = F.conv2d(x_mul, filt).view(B, 2, hs, ws)
    gx = torch.clamp(allf[:, 0:1] / p * 2 - 1, min=-1, max=1)
    gy = torch.clamp(allf[:, 1:2] / p * 2 - 1, min=-1, max=1)
    return torch.cat((gx, gy), 1).permute(0, 2, 3, 1)


# TRAIN.def_saliency_pad_mode -> F.pad mode (models/models.py:819-825: nn.ReplicationPad2d / F.pad 'reflect' / F.pad 'constant')
PAD_MODES = {"replication": "replicate", "reflect": "reflect", "zero": "constant"}


def pad_saliency(xs: torch.Tensor, radius: int, pad_mode: str = "replication") -> torch.Tensor:
    """models/models.py:819-825 with padding_size_x == padding_size_y == radius (gaussian_ap 0.0 on a square saliency map, :496-500)."""
    return F.pad(xs, (radius,) * 4, mode=PAD_MODES[pad_mode])


def create_grid_f64(xs: torch.Tensor, radius: int, pad_mode: str = "replication") -> torch.Tensor:
    """fp64 evaluation of the same formula with the separable Gaussian (accuracy yardstick;
    SURVEY.md §7: the reference's own fp32 result is ~1.75e-5 away from this)."""
    B, _, hs, ws = xs.shape
    g = torch.from_numpy(gaussian_1d(2 * radius + 1, radius))
    x = pad_saliency(xs.double(), radius, pad_mode)[:, 0]
    ci = (torch.arange(ws + 2 * radius, dtype=torch.float64) - radius) / (ws - 1.0)
    ri = (torch.arange(hs + 2 * radius, dtype=torch.float64) - radius) / (hs - 1.0)

    def sep(t):     # valid separable correlation
        t = F.conv2d(t[:, None], g.view(1, 1, 1, -1))
        return F.conv2d(t, g.view(1, 1, -1, 1))[:, 0]
    p = sep(x)
    ax = sep(x * ci[None, None, :])
    ay = sep(x * ri[None, :, None])
    gx = torch.clamp(ax / p * 2 - 1, -1, 1)
    gy = torch.clamp(ay / p * 2 - 1, -1, 1)
    return torch.stack((gx, gy), -1)


def inverse_index_maps(grid: torch.Tensor, H: int, W: int):
    """models/models.py:640-655 -- truncated integer target coordinates (u=col, v=row) of every grid
    point in the full-resolution frame, and the NaN mask of never-hit pixels."""
    u = (((grid[..., 0] + 1) / 2) * (W - 1)).int().long()
    v = (((grid[..., 1] + 1) / 2) * (H - 1)).int().long()
    B = grid.shape[0]
    hit = torch.zeros(B, H, W, dtype=torch.bool)
    hit[torch.arange(B)[:, None], v.view(B, -1), u.view(B, -1)] = True
    return u, v, ~hit


# --------------------------------------------------------------------------------------------
# A9: edge loss; A16-A19: ground truth, losses, accuracies
# --------------------------------------------------------------------------------------------
def edge_loss(xs: torch.Tensor, y: torch.Tensor, hs: int, ws: int, scale: float) -> torch.Tensor:
    """models/models.py:730,889-891,898 -- 0.05*MSE(minmax(xs), minmax(area_pool(y)))*scale with
    whole-batch min/max."""
    t = F.interpolate(y, size=(hs, ws), mode="area")
    a = (xs - xs.min()) / (xs.max() - xs.min())
    b = (t - t.min()) / (t.max() - t.min())
    return 0.05 * F.mse_loss(a, b) * scale


def compose_gt(label: torch.Tensor, cls: torch.Tensor, bg: int = 50) -> torch.Tensor:
    """models/models.py:967-968 -- gt = label*cls + (1-label)*50 (int64)."""
    return label * cls[:, :, None] + (1 - label) * bg


def focal_loss(pred: torch.Tensor, gt: torch.Tensor, gamma: float = 5.0) -> torch.Tensor:
    """models/models.py:87-120 -- pt detached (:109)."""
    C = pred.shape[1]
    z = pred.permute(0, 2, 3, 1).reshape(-1, C)
    logpt = F.log_softmax(z, dim=1).gather(1, gt.reshape(-1, 1)).view(-1)
    pt = logpt.detach().exp()
    return (-1 * (1 - pt) ** gamma * logpt).mean()


def dice_loss_multiclass(pred: torch.Tensor, gt: torch.Tensor, eps: float = 1e-7) -> torch.Tensor:
    """pytorch_toolbelt 0.8.0 DiceLoss(mode='multiclass', from_logits=True, smooth=0, eps=1e-7),
    restated from the published algorithm (package absent; call site models/models.py:482,1059):
    p = exp(log_softmax); per class over (batch, pixels): 1 - 2*sum(p*t)/max(sum(p+t), eps), zeroed
    for classes absent from gt, mean over all classes."""
    B, C = pred.shape[:2]
    p = pred.log_softmax(dim=1).exp().view(B, C, -1)
    t = F.one_hot(gt.view(B, -1), C).permute(0, 2, 1).type_as(p)
    inter = (p * t).sum((0, 2))
    card = (p + t).sum((0, 2))
    loss = 1.0 - (2.0 * inter) / card.clamp_min(eps)
    loss = loss * (t.sum((0, 2)) > 0).to(loss.dtype)
    return loss.mean()


def accuracies(pred: torch.Tensor, gt: torch.Tensor, bg: int = 50):
    """models/models.py:378-474 -- the four per-image IoU-style scores, averaged over the batch."""
    B = pred.shape[0]
    preds = pred.argmax(1)
    out = [0.0, 0.0, 0.0, 0.0]
    for i in range(B):
        p, g = preds[i], gt[i]
        vg, vp = (g < bg), (p < bg)
        bgg, bgp = (g == bg), (p == bg)
        union_fg = (vg | vp).sum().float() + 1e-10
        union_bg = (bgg | bgp).sum().float() + 1e-10
        cls_fg = (vg & (p == g)).sum().float()
        bin_fg = (vg & (vg == vp)).sum().float()
        cls_bg = (bgg & (p == g)).sum().float()
        bin_bg = (bgg & (bgg == bgp)).sum().float()
        out[0] += cls_fg / union_fg
        out[1] += bin_fg / union_fg
        out[2] += 0.5 * cls_fg / union_fg + 0.5 * cls_bg / union_bg
        out[3] += 0.5 * bin_fg / union_fg + 0.5 * bin_bg / union_bg
    return tuple(o / B for o in out)


# --------------------------------------------------------------------------------------------
# A2 + forward: the whole module
# --------------------------------------------------------------------------------------------
class OracleDeformSeg(nn.Module):
    """models/models.py:476-1094 under the effective LVIS-50 configuration:
    joint loss on, loss at low resolution, no upsample, replication pad, learned sampling.  Two off-default settings are restated
    too: pad_mode = TRAIN.def_saliency_pad_mode (:819-825) and uniform = (MODEL.uniform_sample != '') (:816-818)."""

    def __init__(self, hs=80, ws=80, radius=45, num_class=51, fc_dim=960, edge_scale=100.0, pad_mode="replication", uniform=False):
        super().__init__()
        self.hs, self.ws, self.radius, self.edge_scale = hs, ws, radius, edge_scale
        self.pad_mode, self.uniform = pad_mode, uniform
        self.localization = OracleFovSim()
        self.net_compress = OracleCompress()
        self.encoder = OracleHRNet()
        self.decoder = OracleC1(num_class, fc_dim)
        k = 2 * radius + 1
        self.filter = nn.Conv2d(1, 1, (k, k), bias=False)
        with torch.no_grad():
            self.filter.weight[0, 0] = torch.from_numpy(make_gaussian(k, radius)).float()
        self.filter.weight.requires_grad_(False)
        self.register_buffer("P_basis", p_basis(hs, ws, radius, radius), persistent=False)

    # stage functions, exposed separately so every stage can be checked in isolation
    def saliency(self, x, focus):
        x_low = lowres_input(x, focus, self.hs, self.ws)
        s = self.net_compress(self.localization(x_low))
        B = s.shape[0]
        return F.softmax(s.view(B, -1), dim=1).view(B, 1, self.hs, self.ws), x_low

    def grid_from_saliency(self, xs):
        if self.uniform:
            xs = xs * 0 + 1.0 / (self.hs * self.ws)                     # models.py:818 (the edge loss keeps the learned map, :726)
        xs_hm = pad_saliency(xs, self.radius, self.pad_mode)           # models.py:819-825
        return create_grid(xs_hm, self.filter.weight, self.P_basis, self.hs, self.ws)

    def forward(self, feed: Dict[str, torch.Tensor], is_inference=False, drop_fn: Optional[DropFn] = None,
                return_intermediates=False, upsample=False):
        x, y = feed["img_data"], feed["seg_label"]
        y_full = y
        xs, x_low = self.saliency(x, feed["focus_point"])
        grid = self.grid_from_saliency(xs)
        e_loss = edge_loss(xs, y, self.hs, self.ws, self.edge_scale)
        y_s = F.grid_sample(y.float(), grid, align_corners=False).squeeze(1)       # :880
        x_s = F.grid_sample(x, grid, align_corners=False)                          # :909
        feat = self.encoder(x_s, return_feature_maps=True, drop_fn=drop_fn)
        pred = self.decoder(feat)
        label = y_s.long()                                                         # :951
        feed["seg_label"] = label
        gt = compose_gt(label, feed["cls_label"])
        loss = dice_loss_multiclass(pred, gt) + focal_loss(pred, gt) + e_loss      # :1057-1069
        acc = accuracies(pred, gt)
        if upsample:
            # MODEL.upsample (models/models.py:869-873,933-940,1074-1083): accuracies at full resolution on the prediction warped
            # back through the inverse grid with nearest hole filling, against the original label; the loss is unchanged
            with torch.no_grad():
                H, W = y_full.shape[-2:]
                pred_full, _ = unwarp_nearest_ref(pred.detach(), grid.detach(), H, W)
                y_hs = y_full.reshape(y_full.shape[0], H, W).long()
                acc = accuracies(pred_full, compose_gt(y_hs, feed["cls_label"]))
        inter = dict(x_low=x_low, xs=xs, grid=grid, x_sampled=x_s, label=label, feat=feat[0], pred=pred, gt=gt)
        if return_intermediates:
            return loss, acc, e_loss, inter
        if is_inference:
            return loss, acc[0], e_loss, acc[1], acc[2], acc[3]
        return loss, acc[0], e_loss


# --------------------------------------------------------------------------------------------
# A20: optimisers + LR schedule
# --------------------------------------------------------------------------------------------
def lr_for_epoch(epoch: int, lr_mult: float = 0.001, pretrain: int = 100) -> float:
    """train_deform_semantic.py:328-350 with deform_pretrain_bol=True, scale_by_iter=False:
    lr = lr_mult * 0.1 * 0.1**(epoch // 100) for all four optimisers."""
    return lr_mult * (0.1 * 0.1 ** (epoch // pretrain))


def make_optimizers(m: OracleDeformSeg, weight_decay=1e-4, lr=2e-5):
    """train_deform_semantic.py:271-288 -- four Adam(weight_decay=1e-4) over enc/dec/sal/comp."""
    return [torch.optim.Adam(n.parameters(), lr=lr, weight_decay=weight_decay)
            for n in (m.encoder, m.decoder, m.localization, m.net_compress)]


# --------------------------------------------------------------------------------------------
# the dropout-mask hash shared with the HIP kernels (integer work, numpy)
# --------------------------------------------------------------------------------------------
def dropout_keep_mask_nhwc(n_elem: int, key: int, p: float) -> np.ndarray:
    """keep[e] for linear NHWC element index e, identical to `fs_dropout_keep` in csrc/common.h: one hash per element PAIR,
    h = fmix32((e >> 1) * 0x9E3779B1 + key); element 2i compares the low 16 bits of h, element 2i + 1 the high 16 bits, with the 16-bit
    threshold floor(p * 2**32) >> 16."""
    e = np.arange(n_elem, dtype=np.uint64)
    odd = (e & np.uint64(1)).astype(bool)
    h = ((e >> np.uint64(1)) * np.uint64(0x9E3779B1) + np.uint64(key)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    thresh = np.uint64(int(math.floor(p * 4294967296.0)) >> 16)
    return np.where(odd, h >> np.uint64(16), h & np.uint64(0xFFFF)) >= thresh


def layer_key(seed: int, layer_id: int) -> int:
    """Per-layer 32-bit key, identical to `fs_layer_key` in the host code."""
    x = (seed * 0x9E3779B1 + layer_id * 0x85EBCA6B + 0x27D4EB2F) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    x = (x * 0x297A2D39) & 0xFFFFFFFF
    x ^= x >> 15
    return x


# ----------------------------------------------------------------------------------------------
# input pipeline step
# ----------------------------------------------------------------------------------------------
def ingest_sample_ref(img_u8_hwc, mask_u8_hw, pads, focus, frame, cls):
    """DynamicFocus/e_preprocess_scripts/dataset.py:127-142 for one decoded LVIS sample: torchvision ToTensor on a uint8 HWC
    image is `permute(2,0,1).float().div(255)` (transforms/functional.py:to_tensor), then F.pad with zeros by
    (left, right, top, bottom); the mask is cast to float and padded alike; F_2 = (idx_H/HC, idx_W/WC); cls int64."""
    import torch.nn.functional as F
    x = torch.as_tensor(img_u8_hwc).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    y = torch.as_tensor(mask_u8_hw)[None].to(torch.float32)
    x = F.pad(x, tuple(pads))
    y = F.pad(y, tuple(pads)).to(torch.float32)
    f2 = torch.tensor([focus[0] / frame[0], focus[1] / frame[1]], dtype=torch.float32)
    return x, f2, y, torch.tensor([cls], dtype=torch.int64)


# ----------------------------------------------------------------------------------------------
# inverse warp + nearest hole filling
# ----------------------------------------------------------------------------------------------
def inverse_grid_ref(grid: torch.Tensor, Hs: int, Ws: int):
    """models/models.py:639-655.  Several grid points can claim one full-resolution pixel; the reference resolves that with two
    ATen index_put_ calls whose winner depends on the CPU thread schedule (it can even differ between the x and the y channel).
    The restatement fixes the rule a sequential index_put_ gives -- the LAST claimant wins, for both channels -- via numpy's
    advanced assignment (documented: with repeated indices the last value is assigned)."""
    B, h, w, _ = grid.shape
    u = (((grid[..., 0] + 1) / 2) * (Ws - 1)).int().long().view(B, -1).numpy()
    v = (((grid[..., 1] + 1) / 2) * (Hs - 1)).int().long().view(B, -1).numpy()
    x_cor = np.tile(np.arange(w, dtype=np.float32)[None, :], (h, 1)).reshape(-1)
    y_cor = np.tile(np.arange(h, dtype=np.float32)[:, None], (1, w)).reshape(-1)
    inv = np.full((2, B, Hs, Ws), np.nan, dtype=np.float32)
    for b in range(B):
        inv[0, b][v[b], u[b]] = x_cor
        inv[1, b][v[b], u[b]] = y_cor
    inv = torch.from_numpy(inv)
    inv[0] = inv[0] / w * 2 - 1
    inv[1] = inv[1] / h * 2 - 1
    return inv.permute(1, 2, 3, 0)


def unwarp_nearest_ref(pred: torch.Tensor, grid: torch.Tensor, Hs: int, Ws: int):
    """models/models.py:930-940 with rev_deform_interp='nearest': grid_sample through the inverse grid (NaN -> 0), then every
    hole takes the value of its Euclidean-nearest sampled pixel.  The reference asks scipy's NearestNDInterpolator, whose
    choice among equidistant neighbours is unspecified; this restatement takes the smallest (row, col), and
    `nearest_distance_map` lets tests check optimality independently of the tie rule."""
    inv = inverse_grid_ref(grid, Hs, Ws)
    hole = torch.isnan(inv[..., 0])
    out = F.grid_sample(pred, torch.nan_to_num(inv, nan=0.0), align_corners=False)
    B = pred.shape[0]
    for b in range(B):
        ys, xs = torch.where(~hole[b])
        hy, hx = torch.where(hole[b])
        if len(ys) == 0 or len(hy) == 0:
            continue
        d = (hy[:, None] - ys[None, :]) ** 2 + (hx[:, None] - xs[None, :]) ** 2        # valid pixels come in (row, col) order
        j = d.argmin(dim=1)                                                              # first minimum = smallest (row, col)
        out[b][:, hy, hx] = out[b][:, ys[j], xs[j]]
    return out, hole


def nearest_distance_map(hole: torch.Tensor):
    """Squared distance of every pixel to the nearest non-hole pixel of its image (brute force, small inputs)."""
    B, H, W = hole.shape
    out = torch.zeros(B, H, W, dtype=torch.long)
    for b in range(B):
        ys, xs = torch.where(~hole[b])
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        d = (yy.reshape(-1, 1) - ys[None]) ** 2 + (xx.reshape(-1, 1) - xs[None]) ** 2
        out[b] = d.min(dim=1).values.view(H, W)
    return out
