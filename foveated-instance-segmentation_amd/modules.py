"""nn.Module tree with the reference's attribute names / state_dict keys, computing through the HIP ops.

Mirrors: saliency_network.py:302-333 (FovSimModule), models/models.py:360-372 (CompressNet),
models/hrnetv2_nodownsp.py:32-454 (BasicBlock, Bottleneck, HighResolutionModule, HRNetV2),
models/model_utils.py:224-309 (ResidualBlock, ResNet, C1), lib/nn/modules/batchnorm.py:38-61.
Inside the encoder/decoder, activations are NHWC tensors (B,H,W,C); the plugin boundaries take and
return the reference's logical NCHW shapes (as channels-last views, no copy).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_RELU6


# ----------------------------------------------------------------------------------------------
# parameter holders
# ----------------------------------------------------------------------------------------------
class HipConv2d(nn.Module):
    """Holds `weight` (logical (Cout,Cin,R,S), RSCK storage) and optional `bias`."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, dilation=1):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding, self.dilation = cin, cout, k, stride, padding, dilation
        w = ops.new_rsck_weight(cout, cin, k, k)
        nn.init.kaiming_normal_(w)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def extra_repr(self):
        return f"{self.cin}, {self.cout}, k={self.k}, s={self.stride}, p={self.padding}, bias={self.bias is not None}"


class HipBatchNorm2d(nn.Module):
    """BatchNorm parameter/buffer holder.  sync_extras=True adds the three extra buffers the
    reference's SynchronizedBatchNorm2d keeps in its state_dict (batchnorm.py:50-54)."""

    def __init__(self, c, momentum=0.1, sync_extras=True):
        super().__init__()
        self.num_features, self.momentum, self.eps = c, momentum, ops.BN_EPS
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        if sync_extras:
            self.register_buffer("_tmp_running_mean", torch.zeros(c))
            self.register_buffer("_tmp_running_var", torch.ones(c))
            self.register_buffer("_running_iter", torch.ones(1))
        self._pending_batches = 0
        self.register_state_dict_pre_hook(lambda m, prefix, keep_vars: m._flush_batches())

    def _flush_batches(self):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0


class _NbtCounter:
    """Stands in for the num_batches_tracked tensor inside ops.ConvBnAct (host-side count)."""

    def __init__(self, bn):
        self.bn = bn

    def add_(self, n):
        self.bn._pending_batches += n


def conv_bn_act(x, conv: HipConv2d, bn: HipBatchNorm2d, act, res=None, drop_p=0.0, layer_id=0, stride=None, phase=1):
    """phase = d > 1: x (and res) are ALREADY in the phase domain of dilation d (_space_to_batch(., d), done once by the caller for a run of
    layers): a 3x3 layer of dilation d runs as the plain 3x3 / pad 1 layer it is there, a 1x1 layer is unchanged; the result stays in the domain."""
    training = bn.training
    meta = dict(stride=conv.stride if stride is None else stride, pad=conv.padding, dil=conv.dilation, act=act, training=training, momentum=bn.momentum,
                drop_p=drop_p, drop_key=ops.DropoutState.key(layer_id) if (training and drop_p > 0) else 0,
                running_mean=bn.running_mean, running_var=bn.running_var, num_batches_tracked=_NbtCounter(bn),
                grad_enabled=torch.is_grad_enabled())
    x, w = ops.pad_in_channels(x, conv.weight)
    d = meta["dil"]
    if phase > 1:
        assert meta["stride"] == 1 and not (training and drop_p > 0) and ((w.shape[2] == 1 and meta["pad"] == 0) or (w.shape[2] == 3 and d == phase and meta["pad"] == d))
        z = ops.ConvBnAct.apply(x, w, conv.bias, bn.weight, bn.bias, res, dict(meta, dil=1, pad=1 if w.shape[2] == 3 else 0))
    elif (SPACE_TO_BATCH_DILATED and d > 1 and w.shape[2] == 3 and w.shape[3] == 3 and meta["stride"] == 1 and meta["pad"] == d
            and d >= x.shape[1] and d >= x.shape[2]):
        # ASPP rates of 12 / 24 / 36 on a 10 x 10 map: every tap but the centre reads zero padding only, so the layer IS the 1 x 1 conv of its
        # centre tap (and the other taps' gradients are exactly zero)
        z = ops.ConvBnAct.apply(x, ops.CenterTap.apply(w), conv.bias, bn.weight, bn.bias, res, dict(meta, dil=1, pad=0))
    elif (SPACE_TO_BATCH_DILATED and d > 1 and w.shape[2] == 3 and w.shape[3] == 3 and meta["stride"] == 1 and meta["pad"] == d
            and x.shape[1] % d == 0 and x.shape[2] % d == 0 and not (training and drop_p > 0)):
        # An atrous 3x3 conv (DeepLab layer3 / layer4) = d*d ordinary 3x3 convs, pad 1, on the d*d sub-sampled phase images x[d i + a][d j + b]:
        # the phases become batch entries, so the layer runs on the halo-tiled / F(2,3) kernels instead of the general one (configs[4]:
        # 160 us per launch on 10x10 maps there).  BatchNorm sees the same set of elements; zero padding of a phase image is the
        # dilated conv's zero padding.  The two re-orderings are plain views + one copy each way, and differentiable as such.
        meta2 = dict(meta, dil=1, pad=1)
        z = _batch_to_space(ops.ConvBnAct.apply(_space_to_batch(x, d), w, conv.bias, bn.weight, bn.bias,
                                                None if res is None else _space_to_batch(res, d), meta2), d)
    else:
        z = ops.ConvBnAct.apply(x, w, conv.bias, bn.weight, bn.bias, res, meta)
    if ops.ACT_TRACE is not None and act != ACT_NONE:
        ops.ACT_TRACE.append((bn, act, z))
    return z


SPACE_TO_BATCH_DILATED = os.environ.get("FS_S2B_DILATED", "1") != "0"      # A/B switch, read once


def _space_to_batch(x, d):
    """(B,H,W,C) -> (B*d*d, H/d, W/d, C): entry (b, a, bb) holds x[b][d i + a][d j + bb]."""
    B, H, W, C = x.shape
    return x.reshape(B, H // d, d, W // d, d, C).permute(0, 2, 4, 1, 3, 5).reshape(B * d * d, H // d, W // d, C)


def _batch_to_space(y, d):
    Bd, h, w, C = y.shape
    B = Bd // (d * d)
    return y.reshape(B, d, d, h, w, C).permute(0, 3, 1, 4, 2, 5).reshape(B, h * d, w * d, C)


C1_STASH = os.environ.get("FS_C1_STASH", "1") != "0"      # A/B switch, read once: 0 = `feat` read twice, the autograd engine adds the two gradients
PARALLEL_BRANCHES = True
PARALLEL_FUSE = os.environ.get("FS_PARALLEL_FUSE", "1") != "0"      # fuse rows on the branch streams too (A/B switch, read once)
STREAM_DEPS = os.environ.get("FS_STREAM_DEPS", "1") != "0"          # modules of a stage chained stream by stream, one join per stage (A/B switch)
_SIDE = {}


def _side_streams(device, n):
    key = (device.type, device.index)
    pool = _SIDE.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def to_nhwc(x_nchw: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) logical -> contiguous (B,H,W,C); free when x is already a channels-last view."""
    v = x_nchw.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def to_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    return x_nhwc.permute(0, 3, 1, 2)


def _assign_paths(root: nn.Module, prefix: str = ""):
    for name, m in root.named_modules():
        m._path = (prefix + "." + name).strip(".") if name else prefix


# ----------------------------------------------------------------------------------------------
# saliency net + compress
# ----------------------------------------------------------------------------------------------
class FovSimModule(nn.Module):
    def __init__(self, cfg=None, in_channels=5, out_channels=24):
        super().__init__()
        w = 8 * out_channels
        self.fov_expand_1 = HipConv2d(in_channels, w, 3, 1, 1)
        self.fov_expand_2 = HipConv2d(w, w, 3, 1, 1)
        self.fov_squeeze_1 = HipConv2d(w, out_channels, 3, 1, 1)
        self.norm1 = HipBatchNorm2d(w, 0.1)
        self.norm2 = HipBatchNorm2d(w, 0.1)
        self.norm3 = HipBatchNorm2d(out_channels, 0.1)

    def forward_nhwc(self, x):
        a = conv_bn_act(x, self.fov_expand_1, self.norm1, ACT_RELU6)
        b = conv_bn_act(a, self.fov_expand_2, self.norm2, ACT_RELU6)
        return conv_bn_act(b, self.fov_squeeze_1, self.norm3, ACT_NONE)

    def forward(self, x, reset_grad=True, train_mode=True):
        return to_nchw_view(self.forward_nhwc(to_nhwc(x)))


def fov_simple(cfg=None, pretrained=False, in_channels=5, out_channels=24):
    return FovSimModule(cfg, in_channels, out_channels)


class CompressNet(nn.Module):
    """ReLU -> 1x1 conv (24->1, bias).  `forward` returns the logits like the reference; the module
    pipeline uses `softmax_nhwc`, which fuses the spatial softmax (models/models.py:715-723)."""

    def __init__(self, cfg=None):
        super().__init__()
        cin = 24
        self.conv_last = nn.Conv2d(cin, 1, kernel_size=1)    # parameter holder only (key names/shapes)

    def softmax_nhwc(self, s):
        return ops.CompressSoftmax.apply(s, self.conv_last.weight, self.conv_last.bias)

    def forward(self, x):
        """(B,24,H,W) logical NCHW -> logits (B,1,H,W), as the reference calls it (models/models.py:713)."""
        return ops.Compress.apply(to_nhwc(x), self.conv_last.weight, self.conv_last.bias)


# ----------------------------------------------------------------------------------------------
# HRNetV2 (stride-1 stem)
# ----------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, c):
        super().__init__()
        self.conv1 = HipConv2d(c, c, 3, 1, 1)
        self.bn1 = HipBatchNorm2d(c, 0.1)
        self.conv2 = HipConv2d(c, c, 3, 1, 1)
        self.bn2 = HipBatchNorm2d(c, 0.1)
        self.drop_p = 0.3
        self._path = ""

    def forward(self, x):
        id1 = ops.layer_id_from_name(self._path + ".conv1")
        id2 = ops.layer_id_from_name(self._path + ".conv2")
        xa, xr = ops.fan_out(x, 2)              # conv path + residual: their gradients meet in one HIP add
        o = conv_bn_act(xa, self.conv1, self.bn1, ACT_RELU, drop_p=self.drop_p, layer_id=id1)
        return conv_bn_act(o, self.conv2, self.bn2, ACT_RELU, res=xr, drop_p=self.drop_p, layer_id=id2)


class _ConvBn(nn.Sequential):
    """Sequential(conv, bn[, relu]) container with the reference's numeric child names."""

    def __init__(self, cin, cout, k, stride, relu, bias=False, sync=True, momentum=0.1):
        mods = [HipConv2d(cin, cout, k, stride, k // 2, bias=bias), HipBatchNorm2d(cout, momentum, sync)]
        if relu:
            mods.append(nn.ReLU())
        super().__init__(*mods)
        self.relu = relu

    def forward(self, x, res=None, act=None, stride=None):
        a = (ACT_RELU if self.relu else ACT_NONE) if act is None else act
        return conv_bn_act(x, self[0], self[1], a, res=res, stride=stride)


class _Chain(nn.Sequential):
    def forward(self, x):
        for m in self:
            x = m(x)
        return x


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, down):
        super().__init__()
        self.conv1 = HipConv2d(cin, planes, 1)
        self.bn1 = HipBatchNorm2d(planes, 0.1)
        self.conv2 = HipConv2d(planes, planes, 3, 1, 1)
        self.bn2 = HipBatchNorm2d(planes, 0.1)
        self.conv3 = HipConv2d(planes, planes * 4, 1)
        self.bn3 = HipBatchNorm2d(planes * 4, 0.1)
        self.downsample = _ConvBn(cin, planes * 4, 1, 1, False) if down else None

    def forward(self, x):
        xa, xr = ops.fan_out(x, 2)
        r = xr if self.downsample is None else self.downsample(xr)
        o = conv_bn_act(xa, self.conv1, self.bn1, ACT_RELU)
        o = conv_bn_act(o, self.conv2, self.bn2, ACT_RELU)
        return conv_bn_act(o, self.conv3, self.bn3, ACT_RELU, res=r)


class HighResolutionModule(nn.Module):
    def __init__(self, chans):
        super().__init__()
        n = len(chans)
        self.chans = list(chans)
        self.branches = nn.ModuleList([_Chain(*[BasicBlock(c) for _ in range(4)]) for c in chans])
        rows = []
        for i in range(n):
            row = []
            for j in range(n):
                if j > i:
                    row.append(_ConvBn(chans[j], chans[i], 1, 1, False))
                elif j == i:
                    row.append(None)
                else:
                    row.append(_Chain(*[_ConvBn(chans[j], chans[i] if k == i - j - 1 else chans[j], 3, 2, k != i - j - 1)
                                        for k in range(i - j)]))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)

    def _run_branches(self, xs):
        """The n resolution branches are independent until the fuse, so branches 1.. run on side HIP
        streams: their workgroups fill the tail of each other's launches and the HBM-bound BatchNorm
        passes of one branch overlap the MFMA-bound convolutions of another.  Autograd replays each
        node's backward on the stream its forward ran on, so the backward overlaps the same way."""
        n = len(self.chans)
        if not (PARALLEL_BRANCHES and xs[0].is_cuda):
            return [self.branches[i](xs[i]) for i in range(n)]
        main = torch.cuda.current_stream()
        side = _side_streams(xs[0].device, n - 1)
        fork = torch.cuda.Event()
        fork.record(main)
        outs = [None] * n
        for i in range(1, n):
            s = side[i - 1]
            s.wait_event(fork)
            xs[i].record_stream(s)
            with torch.cuda.stream(s):
                outs[i] = self.branches[i](xs[i])
        outs[0] = self.branches[0](xs[0])
        for i in range(1, n):
            main.wait_stream(side[i - 1])
            outs[i].record_stream(main)
        return outs

    def _fuse_row(self, i, fan, xs):
        n = len(self.chans)
        terms = [fan[j][i] if j == i else self.fuse_layers[i][j](fan[j][i]) for j in range(n)]
        return ops.HrFuse.apply(xs[i].shape[1], xs[i].shape[2], *terms)

    def forward(self, xs):
        n = len(self.chans)
        xs = self._run_branches(xs)
        fan = [ops.fan_out(xs[j], n) for j in range(n)]        # every branch output is read by all n fuse rows
        if PARALLEL_BRANCHES and PARALLEL_FUSE and xs[0].is_cuda and ops.ACT_TRACE is None:
            # Round 5: the n fuse rows are independent of each other as well (row i: its up-path 1x1 convs, its stride-2 down chains, one
            # HrFuse), and most of their launches are small (20x20 / 10x10 maps, BatchNorm finalize kernels): row i runs on the side
            # stream branch i ran on -- the stream the NEXT module's branch i will take its input on -- and the backward follows.
            main = torch.cuda.current_stream()
            side = _side_streams(xs[0].device, n - 1)
            fork = torch.cuda.Event()
            fork.record(main)                                   # behind the fan-out aliases (views: no launch) and _run_branches' joins
            outs = [None] * n
            for i in range(1, n):
                s = side[i - 1]
                s.wait_event(fork)
                for j in range(n):
                    fan[j][i].record_stream(s)
                with torch.cuda.stream(s):
                    outs[i] = self._fuse_row(i, fan, xs)
            outs[0] = self._fuse_row(0, fan, xs)
            for i in range(1, n):
                main.wait_stream(side[i - 1])
                outs[i].record_stream(main)
            return outs
        outs = []
        for i in range(n):
            outs.append(self._fuse_row(i, fan, xs))
            if ops.ACT_TRACE is not None:
                ops.ACT_TRACE.append(((self, i), ACT_RELU, outs[-1]))
        return outs

    def forward_deps(self, xs, ready, defer):
        """forward() with the stream dependencies spelled out instead of two joins on the main stream per module (round 5).  Stream of index
        i: the main stream for i = 0, side stream i - 1 otherwise; branch i, row i and the NEXT module's branch i all run on stream i.
        `ready[i]` (or None): xs[i] is complete on stream i only (the previous module of the stage deferred its join) -- branch i, on the
        same stream, needs no wait then.  Between branches and rows the dependency is all-to-all (every row reads every branch): stream i
        waits for the end-of-branch events of the other streams.  defer: return (outs, per-stream readiness) without joining the main stream."""
        n = len(self.chans)
        main = torch.cuda.current_stream()
        side = _side_streams(xs[0].device, n - 1)
        streams = [main] + list(side)
        fork = None
        bouts, bdone = [None] * n, [None] * n
        for i in range(n - 1, -1, -1):               # side streams first, the main stream's branch last (as _run_branches issues them)
            s = streams[i]
            if i > 0 and (ready is None or ready[i] is None):
                if fork is None:
                    fork = torch.cuda.Event()
                    fork.record(main)
                s.wait_event(fork)
                xs[i].record_stream(s)
            with torch.cuda.stream(s):
                bouts[i] = self.branches[i](xs[i])
                bdone[i] = torch.cuda.Event()
                bdone[i].record(s)
        fan = [ops.fan_out(bouts[j], n) for j in range(n)]
        outs, rdone = [None] * n, [None] * n
        for i in range(n - 1, -1, -1):
            s = streams[i]
            for j in range(n):
                if j != i:
                    s.wait_event(bdone[j])
                    fan[j][i].record_stream(s)
            with torch.cuda.stream(s):
                outs[i] = self._fuse_row(i, fan, bouts)
                if defer and i > 0:
                    rdone[i] = torch.cuda.Event()
                    rdone[i].record(s)
        if defer:
            return outs, rdone
        for i in range(1, n):
            main.wait_stream(side[i - 1])
            outs[i].record_stream(main)
        return outs, None


class _Stage(_Chain):
    """The modules of one HRNet stage (reference child names '0', '1', ...).  With the branch streams on, consecutive modules are chained
    stream by stream: row i of module k and branch i of module k + 1 run on the same side stream, so nothing but the last module of the
    stage joins the main stream (HighResolutionModule.forward_deps)."""

    def forward(self, xs):
        if not (PARALLEL_BRANCHES and PARALLEL_FUSE and STREAM_DEPS and xs[0].is_cuda and ops.ACT_TRACE is None) or len(self) < 2:
            return super().forward(xs)
        ready = None
        for k, m in enumerate(self):
            xs, ready = m.forward_deps(xs, ready, defer=k + 1 < len(self))
        return xs


class HRNetV2(nn.Module):
    WIDTHS = (64, 128, 256, 512)

    def __init__(self, n_class=1000, **kwargs):
        super().__init__()
        W = self.WIDTHS
        self.conv1 = HipConv2d(3, 64, 3, 1, 1)
        self.bn1 = HipBatchNorm2d(64, 0.1)
        self.conv2 = HipConv2d(64, 64, 3, 1, 1)
        self.bn2 = HipBatchNorm2d(64, 0.1)
        self.layer1 = _Chain(Bottleneck(64, 64, True), Bottleneck(256, 64, False), Bottleneck(256, 64, False),
                             Bottleneck(256, 64, False))
        self.transition1 = nn.ModuleList([_ConvBn(256, W[0], 3, 1, True), _Chain(_ConvBn(256, W[1], 3, 2, True))])
        self.stage2 = _Chain(*[HighResolutionModule(W[:2]) for _ in range(1)])
        self.transition2 = nn.ModuleList([None, None, _Chain(_ConvBn(W[1], W[2], 3, 2, True))])
        self.stage3 = _Stage(*[HighResolutionModule(W[:3]) for _ in range(4)])
        self.transition3 = nn.ModuleList([None, None, None, _Chain(_ConvBn(W[2], W[3], 3, 2, True))])
        self.stage4 = _Stage(*[HighResolutionModule(W[:4]) for _ in range(3)])
        _assign_paths(self)

    def forward_nhwc(self, x):
        x = conv_bn_act(x, self.conv1, self.bn1, ACT_RELU)
        x = conv_bn_act(x, self.conv2, self.bn2, ACT_RELU)
        x = self.layer1(x)
        ys = [self.transition1[0](x), self.transition1[1](x)]
        ys = self.stage2(ys)
        ys = [ys[0], ys[1], self.transition2[2](ys[-1])]
        ys = self.stage3(ys)
        ys = [ys[0], ys[1], ys[2], self.transition3[3](ys[-1])]
        ys = self.stage4(ys)
        return ops.UpsampleConcat.apply(*ys)

    def forward(self, x, return_feature_maps=False):
        return [to_nchw_view(self.forward_nhwc(to_nhwc(x)))]


def hrnetv2_nodownsp(pretrained=False, **kwargs):
    return HRNetV2(n_class=1000, **kwargs)


# ----------------------------------------------------------------------------------------------
# C1 head
# ----------------------------------------------------------------------------------------------
class ResidualBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = _ConvBn(cin, cout, 3, stride, True, bias=True, sync=False)
        self.conv2 = _ConvBn(cout, cout, 3, 1, False, bias=True, sync=False)
        self.downsample = _ConvBn(cin, cout, 1, stride, False, bias=True, sync=False)
        self.downsample[0].padding = 0

    def forward(self, x):
        ds = self.downsample[0]
        if ops.FANOUT_SUBSAMPLE and ds.k == 1 and ds.stride > 1 and ds.padding == 0:
            # a strided 1x1 conv reads every stride-th pixel: run it densely on the subsampled input, and let the backward add
            # its gradient into conv1's dX at those pixels instead of materialising a full-size, 15/16-zero tensor for autograd to sum
            x, xs = ops.FanOutSubsample.apply(x, ds.stride)
            r = self.downsample(xs, stride=1)
        else:
            r = self.downsample(x)
        o = self.conv1(x)
        return self.conv2(o, res=r, act=ACT_RELU)


class _Linear(nn.Module):
    """nn.Linear parameter holder whose weight (out,in) is stored (in,out) = RSCK of a 1x1 conv."""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cin, cout).t()
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):           # x (B, cin) -> (B, cout)
        B, cin = x.shape
        w4 = ops.param_view(self.weight, lambda t: t.view(t.shape[0], t.shape[1], 1, 1))
        return ops.ConvBias.apply(x.view(B, 1, 1, cin), w4, self.bias, 1, 0).view(B, -1)


class ResNet(nn.Module):
    def __init__(self, inplanes=960, num_classes=51):
        super().__init__()
        self.layer2 = _Chain(ResidualBlock(inplanes, 512, 4))
        self.layer3 = _Chain(ResidualBlock(512, 512, 2))
        self.fc = _Linear(512, num_classes)

    def forward(self, x):
        x = self.layer3(self.layer2(x))
        if x.shape[1] != 10 or x.shape[2] != 10:
            # the reference hard-codes AvgPool2d((10,10)) -> FC(512) and fails for other sizes
            # (model_utils.py:254-255); global average pooling is the documented divergence.
            pass
        return self.fc(ops.AvgPoolHW.apply(x))


class C1(nn.Module):
    def __init__(self, num_class=150, fc_dim=2048, use_softmax=False):
        super().__init__()
        self.use_softmax = use_softmax
        self.num_class = num_class
        self.cbr = _ConvBn(fc_dim, fc_dim // 4, 3, 1, True, momentum=0.001)
        self.conv_last = nn.Conv2d(fc_dim // 4, 1, 1, 1, 0)     # parameter holder (key names/shapes)
        self.cls_net = ResNet(inplanes=fc_dim, num_classes=num_class)

    def forward_nhwc(self, feat):
        # `feat` has two readers.  Through one fan-out, with the classification branch's gradient handed to the 3x3 conv's bwd-data epilogue
        # (ops.StashGrad; the node is created after the mask branch so that its backward runs first): no 1.57 GB + 1.57 GB add pass at B = 64
        fa, fb = ops.fan_out(feat, 2) if C1_STASH else (feat, feat)
        x = self.cbr(fa)
        m = ops.MaskHead.apply(x, self.conv_last.weight, self.conv_last.bias)      # (B,H,W)
        fan = getattr(fb, "_fs_fan", None)
        if fan is not None:
            fb = ops.StashGrad.apply(fb, fan[0])
        cls = self.cls_net(fb)                                                      # (B,K)
        return ops.PredAssemble.apply(cls, m)                                       # (B,K,H,W) NCHW

    def forward(self, conv_out, segSize=None, res=None):
        if res is not None:
            raise NotImplementedError("C1 `res` input is unused on the FovealSeg path (model_utils.py:295-297)")
        return self.forward_nhwc(to_nhwc(conv_out[-1]))
