"""CPU tests of the host side: C-ABI surface, config, LR schedule, sharding, flat arenas, no-fallback."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import fovealseg
from fovealseg import hip, ops, train

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "fovealseg.h")).read()
    return sorted(set(re.findall(r"\b(?:int|long)\s+(fs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = hip.load()                                  # raises if the .so is missing: no fallback
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fovealseg.h but not exported"
    assert sorted(list(hip.SIGNATURES) + list(hip.HOST_ONLY)) == declared     # the ctypes table binds exactly the header
    nm = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (fs_[a-z0-9_]+)", nm)))
    assert exported == declared


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/fovealseg.h must compile as C99 on its own (no torch, no HIP headers)."""
    src = tmp_path / "t.c"
    src.write_text('#include "fovealseg.h"\nint main(void) { int (*f)(void) = fs_get_conv_precision; return f != 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_cpu_fallback():
    with pytest.raises(hip.HipLibraryError):
        hip.ptr(torch.zeros(4))
    with pytest.raises(hip.HipLibraryError):
        ops.gaze_lowres(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2), 4, 4)


def test_cfg_merge_and_defaults():
    cfg = fovealseg.lvis50_cfg()
    assert cfg.MODEL.gaussian_radius == 45 and tuple(cfg.TRAIN.saliency_input_size) == (80, 80)
    cfg.merge_from_list(["TRAIN.task_input_size", "(64,64)", "MODEL.gaussian_radius", "15", "DIR", "x"])
    assert cfg.TRAIN.task_input_size == (64, 64) and cfg.MODEL.gaussian_radius == 15 and cfg.DIR == "x"


def test_builder_errors_match_reference():
    MB = fovealseg.ModelBuilder
    with pytest.raises(Exception, match="Architecture undefined!"):
        MB.build_encoder("resnet50")
    with pytest.raises(Exception, match="Architecture undefined!"):
        MB.build_decoder("upernet")
    sf = MB.build_encoder("segformer", fc_dim=1024)
    assert len(sf.state_dict()) == 1172 and "segformer.encoder.block.2.39.mlp.dwconv.dwconv.weight" in sf.state_dict()
    enc = MB.build_encoder("deeplab", fc_dim=960)
    assert "deeplab.classifier.0.convs.4.1.weight" in enc.state_dict() and len(enc.state_dict()) == 669


def test_state_dict_keys_match_oracle():
    import fovealseg_oracle as O
    m, _ = train.build_module(fovealseg.lvis50_cfg(), device="cpu")
    o = O.OracleDeformSeg()
    sd, so = m.state_dict(), o.state_dict()
    assert set(sd) == set(so) and len(sd) == 2830
    assert all(sd[k].shape == so[k].shape for k in sd)
    # conv weights: reference logical shape, RSCK storage
    w = m.encoder.conv1.weight
    assert tuple(w.shape) == (64, 3, 3, 3) and w.permute(2, 3, 1, 0).is_contiguous()
    # a reference-format checkpoint loads strictly and round-trips
    fovealseg.weights.apply_name_keyed_init(o)
    m.load_state_dict(o.state_dict(), strict=True)
    assert torch.equal(m.state_dict()["encoder.stage3.1.fuse_layers.2.0.0.0.weight"],
                       o.state_dict()["encoder.stage3.1.fuse_layers.2.0.0.0.weight"])


class _Opt:
    def __init__(self, zoom):
        self.param_groups = [dict(lr=2e-5, lr_mult=0.001, zoom=zoom)]


def test_lr_schedule_g12(golden):
    g = golden("g12_lr")
    cfg = fovealseg.lvis50_cfg()
    opts = [_Opt(False), _Opt(False), _Opt(True), _Opt(True)]
    for row in g["table"]:
        train.adjust_learning_rate(opts, 0, cfg, epoch=int(row[0]))
        assert [o.param_groups[0]["lr"] for o in opts] == list(row[1:])


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(103))
    for world in (1, 2, 8):
        seen = []
        for rank in range(world):
            ref = list(DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=0))
            got = train.shard_indices(len(ds), rank, world, epoch_seed=0, shuffle=True)
            assert got == ref
            seen += got
        assert set(seen) == set(ds)


def test_flat_params_views_and_strides():
    w = torch.nn.Parameter(ops.new_rsck_weight(8, 4, 3, 3).normal_())
    b = torch.nn.Parameter(torch.randn(5))
    w0, b0 = w.detach().clone(), b.detach().clone()
    fp = train.FlatParams([w, b])
    assert fp.numel % 4 == 0 and torch.equal(w, w0) and torch.equal(b, b0)
    assert w.permute(2, 3, 1, 0).is_contiguous() and w.grad.stride() == w.stride()
    w.grad.add_(1.0)
    assert float(fp.grad.sum()) == w.numel()
    fp.zero_grad()
    assert float(fp.grad.abs().sum()) == 0.0 and w.grad.data_ptr() == fp.grad.data_ptr()
    with torch.no_grad():
        fp.data.mul_(2.0)
    assert torch.equal(w, 2 * w0)


def test_dropout_hash_statistics():
    import fovealseg_oracle as O
    keep = O.dropout_keep_mask_nhwc(1 << 20, ops.layer_key(3, 77), 0.3)
    assert abs(keep.mean() - 0.7) < 2e-3
    assert O.layer_key(3, 77) == ops.layer_key(3, 77)
    assert not np.array_equal(keep, O.dropout_keep_mask_nhwc(1 << 20, ops.layer_key(3, 78), 0.3))


def test_pad_weight_channels_and_fanout_subsample_autograd():
    """Host-side autograd glue added for the odd-channel and strided-shortcut layers (pure tensor logic, no kernel)."""
    import torch
    from fovealseg import ops
    g = torch.Generator().manual_seed(3)
    # PadWeightChannels: zero taps for the padding channels, gradient sliced back to the parameter's shape
    w = ops.new_rsck_weight(6, 5, 3, 3)
    w.copy_(torch.randn(6, 5, 3, 3, generator=g))
    w.requires_grad_(True)
    wp = ops.PadWeightChannels.apply(w, 16)
    assert wp.shape == (6, 16, 3, 3) and ops.rsck(wp).is_contiguous()
    assert torch.equal(wp[:, :5].detach(), w.detach()) and float(wp[:, 5:].detach().abs().max()) == 0.0
    cot = torch.randn(6, 16, 3, 3, generator=g)
    (wp * cot).sum().backward()
    assert torch.equal(w.grad, cot[:, :5])
    assert ops.padded_in_channels(3) == 16 and ops.padded_in_channels(5) == 16 and ops.padded_in_channels(18) == 20
    # FanOutSubsample: forward = (x, x[:, ::s, ::s]); backward adds the subsampled gradient at the sampled pixels
    x = torch.randn(2, 9, 10, 4, generator=g, requires_grad=True)
    xf, xs = ops.FanOutSubsample.apply(x, 4)
    assert torch.equal(xf, x) and torch.equal(xs, x[:, ::4, ::4, :])
    gf, gs = torch.randn(xf.shape, generator=g), torch.randn(xs.shape, generator=g)
    ((xf * 1.0 * gf).sum() + (xs * gs).sum()).backward()
    want = gf.clone()
    want[:, ::4, ::4, :] += gs
    assert torch.allclose(x.grad, want)
    # only the subsampled branch used
    x2 = torch.randn(1, 5, 5, 2, generator=g, requires_grad=True)
    _, xs2 = ops.FanOutSubsample.apply(x2, 2)
    xs2.sum().backward()
    want2 = torch.zeros_like(x2)
    want2[:, ::2, ::2, :] = 1.0
    assert torch.equal(x2.grad, want2)


def test_driver_contract_surface():
    """bench.py takes the driver's flags and __graft_entry__ exposes build() / smoke(); nothing here touches a GPU."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    for flag in ("--gpus", "--steps", "--warmup", "--conv-precision"):
        assert flag in out.stdout
    src = open(os.path.join(root, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
                '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key
    assert "/root/reference" not in src                      # nothing on the GPU box may read the reference
    sys.path.insert(0, root)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    assert "gfx950" in entry or "build.py" in entry
