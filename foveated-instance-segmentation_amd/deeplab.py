"""DeepLabV3-ResNet101 encoder behind the reference's `deeplab` plugin (models/deeplab.py:11-49,420-426).

The reference wraps `torchvision.models.segmentation.deeplabv3_resnet101()` (torchvision==0.19.1, not in
the reference tree and not installed here) and replaces classifier[1]/[2]/[4]; this module restates that
architecture from its published definition with the same state_dict keys (`deeplab.backbone.*`,
`deeplab.classifier.*`), computing through the HIP ops.  Parity is pinned only against this repo's own
CPU oracle (oracle/deeplab_oracle.py) -- "parity unpinned" w.r.t. torchvision itself.
"""
import os

import torch.nn as nn

from . import ops
from .modules import HipBatchNorm2d, HipConv2d, conv_bn_act, to_nchw_view, to_nhwc, _assign_paths
from . import modules as M
from .ops import ACT_NONE, ACT_RELU

PHASE_DOMAIN = os.environ.get("FS_PHASE_DOMAIN", "1") != "0"      # A/B switch: 0 = every atrous 3x3 re-orders around itself
# A/B switch: 0 = the block input is read twice and the autograd engine adds the two gradients (one ATen add + bn3's own reduction pass per block)
BLOCK_FANOUT = os.environ.get("FS_DEEPLAB_FANOUT", "1") != "0"


def _bn(c):
    return HipBatchNorm2d(c, momentum=0.1, sync_extras=False)


class _Down(nn.Sequential):
    def __init__(self, cin, cout, stride):
        super().__init__(HipConv2d(cin, cout, 1, stride, 0), _bn(cout))

    def forward(self, x):
        return conv_bn_act(x, self[0], self[1], ACT_NONE)


class TVBottleneck(nn.Module):
    """torchvision ResNet Bottleneck (stride on conv2, dilation = padding of conv2)."""

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=False):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, 1)
        self.bn1 = _bn(planes)
        self.conv2 = HipConv2d(planes, planes, 3, stride, dilation, dilation=dilation)
        self.bn2 = _bn(planes)
        self.conv3 = HipConv2d(planes, planes * 4, 1)
        self.bn3 = _bn(planes * 4)
        self.downsample = _Down(inplanes, planes * 4, stride) if downsample else None

    def forward(self, x, phase=1):
        """phase = d > 1: x is in the phase domain of this block's dilation (modules.conv_bn_act); so is the result."""
        # conv path + residual through one fan-out, as in the HRNet Bottleneck (modules.Bottleneck): the residual gradient joins conv1's dx in
        # its bwd-data epilogue, which also forms the BatchNorm-backward sums of the block in front (no ATen add, no reduction pass for that bn3)
        xa, xr = ops.fan_out(x, 2) if BLOCK_FANOUT else (x, x)
        r = xr if self.downsample is None else conv_bn_act(xr, self.downsample[0], self.downsample[1], ACT_NONE, phase=phase)
        o = conv_bn_act(xa, self.conv1, self.bn1, ACT_RELU, phase=phase)
        o = conv_bn_act(o, self.conv2, self.bn2, ACT_RELU, phase=phase)
        return conv_bn_act(o, self.conv3, self.bn3, ACT_RELU, res=r, phase=phase)

    def phase_of(self, x_shape):
        """The phase domain this block can run in: its dilation d when its 3x3 is an atrous stride-1 layer on a map that d divides, else 1."""
        c = self.conv2
        d = c.dilation
        ok = (d > 1 and c.stride == 1 and c.padding == d and x_shape[1] % d == 0 and x_shape[2] % d == 0 and
              (self.downsample is None or (self.downsample[0].stride == 1 and self.downsample[0].k == 1)))
        return d if ok else 1


class _Seq(nn.Sequential):
    def forward(self, x):
        for m in self:
            x = m(x)
        return x


class ResNet101Dilated(nn.Module):
    """resnet101(replace_stride_with_dilation=[False, True, True]) up to layer4 (output stride 8)."""

    def __init__(self):
        super().__init__()
        self.conv1 = HipConv2d(3, 64, 7, 2, 3)
        self.bn1 = _bn(64)
        self.inplanes, self.dilation = 64, 1
        self.layer1 = self._make_layer(64, 3, 1, False)
        self.layer2 = self._make_layer(128, 4, 2, False)
        self.layer3 = self._make_layer(256, 23, 2, True)
        self.layer4 = self._make_layer(512, 3, 2, True)

    def _make_layer(self, planes, blocks, stride, dilate):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = stride != 1 or self.inplanes != planes * 4
        layers = [TVBottleneck(self.inplanes, planes, stride, prev, down)]
        self.inplanes = planes * 4
        layers += [TVBottleneck(self.inplanes, planes, 1, self.dilation, False) for _ in range(1, blocks)]
        return _Seq(*layers)

    def forward(self, x):
        x = conv_bn_act(x, self.conv1, self.bn1, ACT_RELU)
        x = ops.MaxPool.apply(x, 3, 2, 1)
        x = self.layer2(self.layer1(x))
        if not (PHASE_DOMAIN and M.SPACE_TO_BATCH_DILATED and ops.ACT_TRACE is None):
            return self.layer4(self.layer3(x))
        # Round 5: the atrous blocks of layer3 / layer4 run in the phase domain of their dilation (the d*d sub-sampled phase images as batch
        # entries: modules._space_to_batch) -- and since everything else in a Bottleneck is point-wise in space (1x1 convs, BatchNorm over the
        # same element set, ReLU, the residual add), a RUN of blocks with the same dilation enters the domain once and leaves it once,
        # instead of re-ordering around every 3x3 (88 copy launches per configs[4] step for layer3's 22 dilation-2 blocks)
        cur = 1
        for blk in list(self.layer3) + list(self.layer4):
            shape = (x.shape[0] // (cur * cur), x.shape[1] * cur, x.shape[2] * cur, x.shape[3])
            want = blk.phase_of(shape)
            if want != cur:
                if cur > 1:
                    x = M._batch_to_space(x, cur)
                if want > 1:
                    x = M._space_to_batch(x, want)
                cur = want
            x = blk(x, phase=cur)
        return M._batch_to_space(x, cur) if cur > 1 else x


class _ConvBnRelu(nn.Sequential):
    def __init__(self, cin, cout, k, dilation=1):
        super().__init__(HipConv2d(cin, cout, k, 1, dilation * (k // 2), dilation=dilation), _bn(cout), nn.ReLU())

    def forward(self, x):
        return conv_bn_act(x, self[0], self[1], ACT_RELU)


class _ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), HipConv2d(cin, cout, 1), _bn(cout), nn.ReLU())

    def forward(self, x):
        B, H, W, C = x.shape
        p = ops.AvgPoolHW.apply(x).view(B, 1, 1, C)
        return conv_bn_act(p, self[1], self[2], ACT_RELU)          # (B,1,1,cout); broadcast by the concat


class ASPP(nn.Module):
    def __init__(self, cin, rates=(12, 24, 36), cout=256):
        super().__init__()
        mods = [_ConvBnRelu(cin, cout, 1)] + [_ConvBnRelu(cin, cout, 3, r) for r in rates] + [_ASPPPooling(cin, cout)]
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(HipConv2d(len(mods) * cout, cout, 1), _bn(cout), nn.ReLU(), nn.Dropout(0.5))
        self._path = ""

    def forward(self, x):
        cat = ops.UpsampleConcat.apply(*[m(x) for m in self.convs])   # pooled branch up-sampled (= broadcast)
        y = conv_bn_act(cat, self.project[0], self.project[1], ACT_RELU)
        if self.training and self.project[3].p > 0:
            key = ops.DropoutState.key(ops.layer_id_from_name(self._path + ".project.3"))
            y = ops.Dropout.apply(y, self.project[3].p, key)
        return y


class _Head(nn.Sequential):
    """DeepLabHead with the reference's replacements: [1]=Conv1x1(256,512,bias) [2]=BN(512) [4]=Conv1x1(512,nc,bias)."""

    def __init__(self, num_classes):
        super().__init__(ASPP(2048), HipConv2d(256, 512, 1, bias=True), _bn(512), nn.ReLU(), HipConv2d(512, num_classes, 1, bias=True))

    def forward(self, x):
        x = self[0](x)
        x = conv_bn_act(x, self[1], self[2], ACT_RELU)
        return ops.ConvBias.apply(x, self[4].weight, self[4].bias, 1, 0)


class _DeepLabV3(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.backbone = ResNet101Dilated()
        self.classifier = _Head(num_classes)


class CustomDeepLab(nn.Module):
    def __init__(self, num_input_channels=3, num_classes=1):
        super().__init__()
        self.deeplab = _DeepLabV3(num_classes)
        nn.init.normal_(self.deeplab.classifier[4].weight.data)
        nn.init.normal_(self.deeplab.classifier[1].weight.data)
        _assign_paths(self)

    def forward_nhwc(self, x):
        B, H, W, _ = x.shape
        f = self.deeplab.classifier(self.deeplab.backbone(x))
        return ops.UpsampleTo.apply(f, H, W)

    def forward(self, x, return_feature_maps=False):
        return [to_nchw_view(self.forward_nhwc(to_nhwc(x)))]


def deeplab(pretrained=False, return_feature_maps=False, **kwargs):
    return CustomDeepLab(num_input_channels=3, num_classes=960)
