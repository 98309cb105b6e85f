"""CPU ORACLE for the DeepLab encoder plugin -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `torchvision==0.19.1` `deeplabv3_resnet101()` (third-party; pinned in the reference's
requirements.txt, absent from /root/reference and not installed here) plus the reference's replacements
(models/deeplab.py:27-31) from the published architecture: ResNet-101 with
replace_stride_with_dilation=[False, True, True], ASPP rates (12,24,36) + image-pooling branch,
projection with Dropout(0.5), classifier[1]=Conv1x1(256,512)+BN, classifier[4]=Conv1x1(512,960),
bilinear up-sampling to the input size.  PARITY UNPINNED: the reference holds no fixture at this
boundary and torchvision cannot be imported, so this oracle is checked against nothing but itself.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from fovealseg_oracle import ORelu, _relu, _site      # activation-replay test instrument (see fovealseg_oracle.ACT_REPLAY)


class _Bottle(nn.Module):
    def __init__(self, inplanes, planes, stride, dilation, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4)) if down else None

    def forward(self, x):
        o = _relu(_site(self, "bn1"), self.bn1(self.conv1(x)))
        o = _relu(_site(self, "bn2"), self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        return _relu(_site(self, "bn3"), o + (x if self.downsample is None else self.downsample(x)))


class _Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.inplanes, self.dilation = 64, 1
        self.layer1 = self._layer(64, 3, 1, False)
        self.layer2 = self._layer(128, 4, 2, False)
        self.layer3 = self._layer(256, 23, 2, True)
        self.layer4 = self._layer(512, 3, 2, True)

    def _layer(self, planes, blocks, stride, dilate):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = stride != 1 or self.inplanes != planes * 4
        mods = [_Bottle(self.inplanes, planes, stride, prev, down)]
        self.inplanes = planes * 4
        mods += [_Bottle(self.inplanes, planes, 1, self.dilation, False) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = F.max_pool2d(_relu(_site(self, "bn1"), self.bn1(self.conv1(x))), 3, 2, 1)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class _ASPP(nn.Module):
    def __init__(self, cin, rates=(12, 24, 36), cout=256):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), ORelu())]
        mods += [nn.Sequential(nn.Conv2d(cin, cout, 3, padding=r, dilation=r, bias=False), nn.BatchNorm2d(cout), ORelu()) for r in rates]
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False), nn.BatchNorm2d(cout), ORelu()))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(len(mods) * cout, cout, 1, bias=False), nn.BatchNorm2d(cout), ORelu(), nn.Dropout(0.5))

    def forward(self, x, drop_fn=None):
        outs = [m(x) for m in self.convs[:-1]]
        outs.append(F.interpolate(self.convs[-1](x), size=x.shape[-2:], mode="bilinear", align_corners=False))
        y = self.project[2](self.project[1](self.project[0](torch.cat(outs, 1))))
        if drop_fn is not None:
            return drop_fn(y)
        return self.project[3](y)


class _Net(nn.Module):
    def __init__(self, nc):
        super().__init__()
        self.backbone = _Backbone()
        self.classifier = nn.Sequential(_ASPP(2048), nn.Conv2d(256, 512, 1), nn.BatchNorm2d(512), ORelu(), nn.Conv2d(512, nc, 1))


class OracleDeepLab(nn.Module):
    def __init__(self, num_classes=960):
        super().__init__()
        self.deeplab = _Net(num_classes)

    def forward(self, x, return_feature_maps=False, drop_fn=None):
        c = self.deeplab.classifier
        f = c[0](self.deeplab.backbone(x), drop_fn)
        f = c[4](c[3](c[2](c[1](f))))
        return [F.interpolate(f, size=x.shape[-2:], mode="bilinear", align_corners=False)]
