"""CPU ORACLE for the SegFormer encoder plugin -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates the Mix-Transformer encoder of `transformers==4.46.2` `SegformerForSemanticSegmentation` (third-party;
pinned in the reference's requirements.txt, absent from /root/reference; call sites models/segformer.py:2,9-11,
33-37,88-100) with the reference's configuration (models/segformer.py:88-99) and its 4-way up-sample + concat
(models/segformer.py:46-53), using the 4.46.2 state_dict key names.  Pinned by tests/golden/g13_segformer.npz
against `transformers 5.15.0` `SegformerModel` (the installed version; tests/golden/make_segformer_golden.py) --
i.e. "transformers 5.15.0 behaviour"; PARITY UNPINNED w.r.t. 4.46.2 itself.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

DEPTHS, HIDDEN, STRIDES, PATCH = (3, 6, 40, 3), (64, 128, 320, 512), (1, 2, 2, 2), (7, 3, 3, 3)
SR, HEADS, MLP_RATIO, DROP_PATH, HIDDEN_DROPOUT, ATTN_DROPOUT, LN_EPS = (8, 4, 2, 1), (1, 2, 5, 8), 4, 0.1, 0.3, 0.2, 1e-6


class Hooks:
    """Randomness injection: each returns the tensor to use; defaults = torch's own RNG semantics."""

    def dropout(self, path, x, p, training):
        return F.dropout(x, p, training)

    def attn_dropout(self, path, probs, p, training):
        return F.dropout(probs, p, training)

    def drop_path(self, path, y, p, training):
        if not training or p == 0.0:
            return y
        keep = torch.empty(y.shape[0], *([1] * (y.dim() - 1))).bernoulli_(1 - p)
        return y * keep / (1 - p)


class _Attn(nn.Module):
    def __init__(self, hidden, heads, sr):
        super().__init__()
        self.heads, self.sr_ratio = heads, sr
        self.query, self.key, self.value = nn.Linear(hidden, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, hidden)
        if sr > 1:
            self.sr = nn.Conv2d(hidden, hidden, sr, sr)
            self.layer_norm = nn.LayerNorm(hidden, eps=LN_EPS)

    def forward(self, x, h, w, hooks, path, training):
        B, N, C = x.shape
        d = C // self.heads
        q = self.query(x).view(B, N, self.heads, d).transpose(1, 2)
        kv = x
        if self.sr_ratio > 1:
            kv = self.sr(x.transpose(1, 2).reshape(B, C, h, w)).reshape(B, C, -1).transpose(1, 2)
            kv = self.layer_norm(kv)
        k = self.key(kv).view(B, -1, self.heads, d).transpose(1, 2)
        v = self.value(kv).view(B, -1, self.heads, d).transpose(1, 2)
        probs = torch.softmax(q @ k.transpose(-1, -2) / d ** 0.5, dim=-1)
        probs = hooks.attn_dropout(path + ".dropout", probs, ATTN_DROPOUT, training)
        return (probs @ v).transpose(1, 2).reshape(B, N, C)


class _SelfOut(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = nn.Linear(hidden, hidden)


class _Attention(nn.Module):
    def __init__(self, hidden, heads, sr):
        super().__init__()
        self.self = _Attn(hidden, heads, sr)
        self.output = _SelfOut(hidden)


class _DW(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class _FFN(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense1 = nn.Linear(hidden, hidden * MLP_RATIO)
        self.dwconv = _DW(hidden * MLP_RATIO)
        self.dense2 = nn.Linear(hidden * MLP_RATIO, hidden)


class _Layer(nn.Module):
    def __init__(self, hidden, heads, sr, dp):
        super().__init__()
        self.layer_norm_1 = nn.LayerNorm(hidden, eps=LN_EPS)
        self.attention = _Attention(hidden, heads, sr)
        self.layer_norm_2 = nn.LayerNorm(hidden, eps=LN_EPS)
        self.mlp = _FFN(hidden)
        self.dp = float(dp)

    def forward(self, x, h, w, hooks, path, training):
        a = self.attention.self(self.layer_norm_1(x), h, w, hooks, path + ".attention.self", training)
        a = hooks.dropout(path + ".attention.output.dropout", self.attention.output.dense(a), HIDDEN_DROPOUT, training)
        x = x + hooks.drop_path(path + ".drop_path1", a, self.dp, training)
        m = self.mlp.dense1(self.layer_norm_2(x))
        B, N, C4 = m.shape
        m = self.mlp.dwconv.dwconv(m.transpose(1, 2).reshape(B, C4, h, w)).flatten(2).transpose(1, 2)
        m = hooks.dropout(path + ".mlp.dropout1", F.gelu(m), HIDDEN_DROPOUT, training)
        m = hooks.dropout(path + ".mlp.dropout2", self.mlp.dense2(m), HIDDEN_DROPOUT, training)
        return x + hooks.drop_path(path + ".drop_path2", m, self.dp, training)


class _Embed(nn.Module):
    def __init__(self, patch, stride, cin, cout):
        super().__init__()
        self.proj = nn.Conv2d(cin, cout, patch, stride, patch // 2)
        self.layer_norm = nn.LayerNorm(cout, eps=LN_EPS)


class _Encoder(nn.Module):
    def __init__(self, num_input=3):
        super().__init__()
        dpr = np.linspace(0, DROP_PATH, sum(DEPTHS)).tolist()
        self.patch_embeddings = nn.ModuleList([_Embed(PATCH[i], STRIDES[i], num_input if i == 0 else HIDDEN[i - 1], HIDDEN[i]) for i in range(4)])
        blocks, cur = [], 0
        for i in range(4):
            blocks.append(nn.ModuleList([_Layer(HIDDEN[i], HEADS[i], SR[i], dpr[cur + j]) for j in range(DEPTHS[i])]))
            cur += DEPTHS[i]
        self.block = nn.ModuleList(blocks)
        self.layer_norm = nn.ModuleList([nn.LayerNorm(HIDDEN[i], eps=LN_EPS) for i in range(4)])


class _Model(nn.Module):
    def __init__(self, num_input):
        super().__init__()
        self.encoder = _Encoder(num_input)


class _Proj(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.proj = nn.Linear(cin, 256)


class _Head(nn.Module):
    def __init__(self, nl):
        super().__init__()
        self.linear_c = nn.ModuleList([_Proj(h) for h in HIDDEN])
        self.linear_fuse = nn.Conv2d(1024, 256, 1, bias=False)
        self.batch_norm = nn.BatchNorm2d(256)
        self.classifier = nn.Conv2d(256, nl, 1)


class OracleSegformer(nn.Module):
    def __init__(self, num_labels=960, num_input=3):
        super().__init__()
        self.segformer = _Model(num_input)
        self.decode_head = _Head(num_labels)

    def stages(self, x, hooks=None):
        hooks = hooks or Hooks()
        enc = self.segformer.encoder
        outs = []
        for i in range(4):
            e = enc.patch_embeddings[i]
            x = e.proj(x)
            B, C, h, w = x.shape
            t = e.layer_norm(x.flatten(2).transpose(1, 2))
            for j, blk in enumerate(enc.block[i]):
                t = blk(t, h, w, hooks, f"segformer.encoder.block.{i}.{j}", self.training)
            t = enc.layer_norm[i](t)
            x = t.reshape(B, h, w, C).permute(0, 3, 1, 2).contiguous()
            outs.append(x)
        return outs

    def forward(self, pixel_values, return_feature_maps=True, hooks=None):
        outs = self.stages(pixel_values, hooks)
        size = outs[0].shape[-2:]
        ups = [outs[0]] + [F.interpolate(o, size=size, mode="bilinear", align_corners=False) for o in outs[1:]]
        return [torch.cat(ups, 1)]


def hf_key(k: str) -> str:
    """Map a 4.46.2-style encoder key of this oracle to transformers 5.15.0 `SegformerModel` naming."""
    import re
    k = k.replace("segformer.encoder.", "")
    k = re.sub(r"^patch_embeddings\.(\d+)\.", r"stages.\1.patch_embeddings.", k)
    k = re.sub(r"^block\.(\d+)\.(\d+)\.", r"stages.\1.blocks.\2.", k)
    k = re.sub(r"^layer_norm\.(\d+)\.", r"stages.\1.layer_norm.", k)
    k = k.replace("layer_norm_1", "layernorm_before").replace("layer_norm_2", "layernorm_after")
    k = k.replace("attention.self.query", "attention.q_proj").replace("attention.self.key", "attention.k_proj")
    k = k.replace("attention.self.value", "attention.v_proj").replace("attention.output.dense", "attention.o_proj")
    k = k.replace("attention.self.sr", "attention.sequence_reduction.sequence_reduction")
    k = k.replace("attention.self.layer_norm", "attention.sequence_reduction.layer_norm")
    k = k.replace("mlp.dense1", "mlp.fc1").replace("mlp.dense2", "mlp.fc2")
    return k
