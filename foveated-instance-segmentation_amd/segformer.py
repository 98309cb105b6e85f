"""SegFormer (Mix-Transformer B5-shaped) encoder behind the reference's `segformer` plugin
(models/segformer.py:9-60,87-105).

The reference subclasses `transformers==4.46.2` `SegformerForSemanticSegmentation` (third-party, not in the
reference tree) with depths (3,6,40,3), hidden sizes (64,128,320,512), strides (1,2,2,2), patch sizes
(7,3,3,3), sequence-reduction ratios (8,4,2,1), heads (1,2,5,8), MLP ratio 4, drop-path 0.1, hidden dropout
0.3, attention dropout 0.2, LayerNorm eps 1e-6, and concatenates the four stage outputs up-sampled to the
stage-1 size (1024 channels).  This module restates that network with the 4.46.2 state_dict key names
(`segformer.encoder.*`, plus the unused `decode_head.*` parameters), computing through the HIP ops.
Parity: against oracle/segformer_oracle.py, which is pinned to transformers 5.15.0's `SegformerModel`
(the version installed here) by tests/golden/g13_segformer.npz -- "parity unpinned" w.r.t. 4.46.2 itself.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .modules import HipConv2d, _assign_paths, to_nchw_view, to_nhwc

DEPTHS = (3, 6, 40, 3)
HIDDEN = (64, 128, 320, 512)
STRIDES = (1, 2, 2, 2)
PATCH = (7, 3, 3, 3)
SR = (8, 4, 2, 1)
HEADS = (1, 2, 5, 8)
MLP_RATIO = 4
DROP_PATH = 0.1
HIDDEN_DROPOUT = 0.3
ATTN_DROPOUT = 0.2
LN_EPS = 1e-6
DECODER_HIDDEN = 256


class HipLinear(nn.Module):
    """nn.Linear holder: logical weight (out,in) stored (in,out) = RSCK of a 1x1 conv; acts on (..., in)."""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cin, cout).t()
        nn.init.trunc_normal_(w, std=0.02)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x, drop_p=0.0, drop_path="", residual=None):
        """drop_p > 0 (training): the nn.Dropout(drop_p) that follows this layer, keyed by its module path, fused into the epilogue.
        residual = (res, droppath_p, droppath_key): the layer ends a residual branch -- returns res + DropPath(Dropout(linear(x))), with the
        add in the GEMM epilogue where that kernel runs the layer (ops.linear_residual), as a separate pass otherwise."""
        cin = x.shape[-1]
        lead = x.shape[:-1]
        w4 = ops.param_view(self.weight, lambda t: t.view(t.shape[0], t.shape[1], 1, 1))
        key = ops.DropoutState.key(ops.layer_id_from_name(drop_path)) if drop_p > 0 else 0
        if residual is not None:
            res, dp_p, dp_key = residual
            out = ops.linear_residual(x, w4, self.bias, res, float(drop_p), key, float(dp_p), int(dp_key))
            if out is not None:
                return out
        y = ops.ConvBias.apply(x.reshape(1, -1, 1, cin), w4, self.bias, 1, 0, float(drop_p), key).view(*lead, -1)
        if residual is not None:
            return ops.ResidualDropPath.apply(res, y, dp_p, dp_key)
        return y


LN_FAN = os.environ.get("FS_LN_FAN", "1") != "0"      # A/B switch: 0 = separate LayerNorm node, the engine adds the residual's gradient


def _ln(mod, x):
    return ops.LayerNorm.apply(x, mod.weight, mod.bias, LN_EPS)


def _drop(x, p, training, path):
    if not training or p <= 0:
        return x
    return ops.Dropout.apply(x, p, ops.DropoutState.key(ops.layer_id_from_name(path)))


class OverlapPatchEmbeddings(nn.Module):
    def __init__(self, patch, stride, cin, cout):
        super().__init__()
        self.proj = HipConv2d(cin, cout, patch, stride, patch // 2, bias=True)
        self.layer_norm = nn.LayerNorm(cout, eps=LN_EPS)

    def forward(self, x):                                   # (B,H,W,Cin) -> (B,h,w,C)
        y = ops.conv_bias_any(x, self.proj.weight, self.proj.bias, self.proj.stride, self.proj.padding)
        return _ln(self.layer_norm, y)


class EfficientSelfAttention(nn.Module):
    def __init__(self, hidden, heads, sr):
        super().__init__()
        self.heads, self.sr_ratio = heads, sr
        self.query, self.key, self.value = HipLinear(hidden, hidden), HipLinear(hidden, hidden), HipLinear(hidden, hidden)
        if sr > 1:
            self.sr = HipConv2d(hidden, hidden, sr, sr, 0, bias=True)
            self.layer_norm = nn.LayerNorm(hidden, eps=LN_EPS)
        self._path = ""

    def forward(self, x):                                   # (B,h,w,C)
        B, h, w, C = x.shape
        q = self.query(x).view(B, h * w, C)
        kv = x
        if self.sr_ratio > 1:
            kv = ops.conv_bias_any(x, self.sr.weight, self.sr.bias, self.sr_ratio, 0)
            kv = _ln(self.layer_norm, kv)
        Nk = kv.shape[1] * kv.shape[2]
        k = self.key(kv).view(B, Nk, C)
        v = self.value(kv).view(B, Nk, C)
        p = ATTN_DROPOUT if self.training else 0.0
        key = ops.DropoutState.key(ops.layer_id_from_name(self._path + ".dropout")) if p > 0 else 0
        return ops.attention(q, k, v, self.heads, p, key).view(B, h, w, C)


class SelfOutput(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = HipLinear(hidden, hidden)
        self._path = ""

    def forward(self, x, residual=None):
        return self.dense(x, HIDDEN_DROPOUT if self.training else 0.0, self._path + ".dropout", residual=residual)


class SegformerAttention(nn.Module):
    def __init__(self, hidden, heads, sr):
        super().__init__()
        self.self = EfficientSelfAttention(hidden, heads, sr)
        self.output = SelfOutput(hidden)

    def forward(self, x, residual=None):
        return self.output(self.self(x), residual=residual)


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)      # parameter holder

    def forward(self, x):
        return ops.DwConv3.apply(x, self.dwconv.weight, self.dwconv.bias)


class MixFFN(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense1 = HipLinear(hidden, hidden * MLP_RATIO)
        self.dwconv = DWConv(hidden * MLP_RATIO)
        self.dense2 = HipLinear(hidden * MLP_RATIO, hidden)
        self._path = ""

    def forward(self, x, residual=None):
        y = self.dwconv(self.dense1(x))
        if self.training and HIDDEN_DROPOUT > 0 and y.numel() < 2 ** 32:
            y = ops.GeluDropout.apply(y, HIDDEN_DROPOUT, ops.DropoutState.key(ops.layer_id_from_name(self._path + ".dropout1")))
        else:
            y = _drop(ops.Gelu.apply(y), HIDDEN_DROPOUT, self.training, self._path + ".dropout1")
        return self.dense2(y, HIDDEN_DROPOUT if self.training else 0.0, self._path + ".dropout2", residual=residual)


class SegformerLayer(nn.Module):
    def __init__(self, hidden, heads, sr, drop_path):
        super().__init__()
        self.layer_norm_1 = nn.LayerNorm(hidden, eps=LN_EPS)
        self.attention = SegformerAttention(hidden, heads, sr)
        self.layer_norm_2 = nn.LayerNorm(hidden, eps=LN_EPS)
        self.mlp = MixFFN(hidden)
        self.drop_path_rate = float(drop_path)
        self._path = ""

    def _dp(self, x, y, tag):
        p = self.drop_path_rate if self.training else 0.0
        key = ops.DropoutState.key(ops.layer_id_from_name(self._path + tag)) if p > 0 else 0
        return ops.ResidualDropPath.apply(x, y, p, key)

    def _res(self, x, tag):
        p = self.drop_path_rate if self.training else 0.0
        return (x, p, ops.DropoutState.key(ops.layer_id_from_name(self._path + tag)) if p > 0 else 0)

    def forward(self, x):
        if LN_FAN:           # x feeds the LayerNorm and the residual add: one node for both, the gradients meet inside the LayerNorm backward
            y, x = ops.LayerNormFan.apply(x, self.layer_norm_1.weight, self.layer_norm_1.bias, LN_EPS)
            x = self.attention(y, residual=self._res(x, ".drop_path1"))      # ... and the residual add is the last linear layer's epilogue
            y, x = ops.LayerNormFan.apply(x, self.layer_norm_2.weight, self.layer_norm_2.bias, LN_EPS)
            return self.mlp(y, residual=self._res(x, ".drop_path2"))
        x = self._dp(x, self.attention(_ln(self.layer_norm_1, x)), ".drop_path1")
        return self._dp(x, self.mlp(_ln(self.layer_norm_2, x)), ".drop_path2")


class SegformerEncoder(nn.Module):
    def __init__(self, num_input=3):
        super().__init__()
        dpr = np.linspace(0, DROP_PATH, sum(DEPTHS)).tolist()
        self.patch_embeddings = nn.ModuleList(
            [OverlapPatchEmbeddings(PATCH[i], STRIDES[i], num_input if i == 0 else HIDDEN[i - 1], HIDDEN[i]) for i in range(4)])
        blocks, cur = [], 0
        for i in range(4):
            blocks.append(nn.ModuleList([SegformerLayer(HIDDEN[i], HEADS[i], SR[i], dpr[cur + j]) for j in range(DEPTHS[i])]))
            cur += DEPTHS[i]
        self.block = nn.ModuleList(blocks)
        self.layer_norm = nn.ModuleList([nn.LayerNorm(HIDDEN[i], eps=LN_EPS) for i in range(4)])

    def forward(self, x):
        outs = []
        for i in range(4):
            x = self.patch_embeddings[i](x)
            for blk in self.block[i]:
                x = blk(x)
            x = _ln(self.layer_norm[i], x)
            outs.append(x)
        return outs


class _SegformerModel(nn.Module):
    def __init__(self, num_input):
        super().__init__()
        self.encoder = SegformerEncoder(num_input)


class _MLPProj(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.proj = nn.Linear(cin, DECODER_HIDDEN)


class _DecodeHead(nn.Module):
    """Parameters of SegformerDecodeHead: present in the reference's state_dict, never used by its forward
    (models/segformer.py:55-59) -- kept only so checkpoints interchange."""

    def __init__(self, num_labels):
        super().__init__()
        self.linear_c = nn.ModuleList([_MLPProj(h) for h in HIDDEN])
        self.linear_fuse = nn.Conv2d(DECODER_HIDDEN * 4, DECODER_HIDDEN, 1, bias=False)
        self.batch_norm = nn.BatchNorm2d(DECODER_HIDDEN)
        self.classifier = nn.Conv2d(DECODER_HIDDEN, num_labels, 1)


class CustomSegformer(nn.Module):
    def __init__(self, num_labels=960, num_input=3):
        super().__init__()
        self.segformer = _SegformerModel(num_input)
        self.decode_head = _DecodeHead(num_labels)
        for p in self.decode_head.parameters():
            p.requires_grad_(False)
        _assign_paths(self)

    def forward_nhwc(self, x):
        return ops.UpsampleConcat.apply(*self.segformer.encoder(x))

    def forward(self, pixel_values, return_feature_maps=True):
        return [to_nchw_view(self.forward_nhwc(to_nhwc(pixel_values)))]


def segformer(pretrained=False, **kwargs):
    return CustomSegformer(num_labels=960, num_input=3)
