"""SURVEY.md §8(f) rows next to the hot path: input-pipeline ingest (f-1), metric reduction (f-2), checkpoint/resume (f-4).
CPU tests cover the oracle restatement, the host logic and the 2-rank reduction; the `gpu` tests compare the HIP ingest
kernel with the oracle bit for bit through the C ABI."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fovealseg
from fovealseg import data, train
from oracle import fovealseg_oracle as O

HAS_GPU = torch.cuda.is_available()


def _sample(seed, H, W, pads, Ci=4):
    g = np.random.default_rng(seed)
    img = g.integers(0, 256, size=(H, W, Ci), dtype=np.uint8)
    mask = (g.random((H, W)) > 0.7).astype(np.uint8)
    l, r, t, b = pads
    return data.Sample(img, mask, pads, focus=(t + H // 2, l + W // 3), frame=(H + t + b, W + l + r), cls=seed % 50)


# ---------------------------------------------------------------------------- f-1 (oracle) ---------------
def test_ingest_oracle_semantics():
    s = _sample(3, 5, 7, (1, 2, 3, 0))
    x, f2, y, cls = O.ingest_sample_ref(s.img.numpy(), s.mask.numpy(), s.pads, s.focus, s.frame, s.cls)
    assert x.shape == (4, 8, 10) and y.shape == (1, 8, 10) and x.dtype == torch.float32 and y.dtype == torch.float32
    inner = x[:, 3:8, 1:8]
    assert torch.equal(inner, torch.from_numpy(s.img.numpy().astype(np.float32) / np.float32(255)).permute(2, 0, 1))
    assert float(x[:, :3].abs().sum()) == 0 and float(x[:, :, :1].abs().sum()) == 0 and float(x[:, :, 8:].abs().sum()) == 0
    assert torch.equal(y[0, 3:8, 1:8], s.mask.float())
    assert cls.dtype == torch.int64 and torch.allclose(f2, torch.tensor([5 / 8, 3 / 10]))


def test_sample_rejects_non_uint8():
    with pytest.raises(TypeError):
        data.Sample(np.zeros((4, 4, 4), np.float32), np.zeros((4, 4), np.uint8), (0, 0, 0, 0), (0, 0), (4, 4), 0)


# ---------------------------------------------------------------------------- f-1 (HIP) ------------------
@pytest.mark.gpu
@pytest.mark.parametrize("channels", [3, 4])
def test_ingest_matches_oracle_bit_exact(channels):
    samples = [_sample(1, 37, 53, (5, 6, 20, 7)), _sample(2, 64, 40, (11, 13, 0, 0)), _sample(3, 60, 64, (0, 0, 1, 3))]
    assert len({s.padded_hw for s in samples}) == 1
    X, F, Y, cls = data.ingest_batch(samples, "cuda", channels=channels)
    for b, s in enumerate(samples):
        x, f2, y, c = O.ingest_sample_ref(s.img.numpy(), s.mask.numpy(), s.pads, s.focus, s.frame, s.cls)
        assert torch.equal(X[b].cpu(), x[:channels])          # u8/255 and zero padding, bit for bit
        assert torch.equal(Y[b].cpu(), y)
        assert torch.equal(F[b].cpu(), f2) and torch.equal(cls[b].cpu(), c)


@pytest.mark.gpu
def test_ingest_rejects_bad_arguments():
    s = _sample(1, 8, 8, (0, 0, 0, 0))
    img = s.img.cuda()
    X = torch.empty(1, 4, 8, 8, device="cuda")
    with pytest.raises(fovealseg.hip.HipLibraryError):      # more output channels than the image has
        fovealseg.hip.call("fs_ingest_sample", img.data_ptr(), None, X.data_ptr(), None, 0, 8, 8, 4, 5, 0, 0, 0, 0)
    with pytest.raises(fovealseg.hip.HipLibraryError):      # Y without a mask
        fovealseg.hip.call("fs_ingest_sample", img.data_ptr(), None, X.data_ptr(), X.data_ptr(), 0, 8, 8, 4, 4, 0, 0, 0, 0)


@pytest.mark.gpu
def test_prefetcher_matches_direct_ingest():
    batches = [[_sample(10 * k + i, 30 + i, 50 - i, (i, 14, 2 * i, 34 - 3 * i)) for i in range(3)] for k in range(4)]
    got = []
    for X, F, Y, cls in data.DevicePrefetcher(batches, "cuda", channels=3):
        got.append((X.clone(), F.clone(), Y.clone(), cls.clone()))
        torch.cuda.current_stream().synchronize()
    assert len(got) == len(batches)
    for samples, (X, F, Y, cls) in zip(batches, got):
        Xr, Fr, Yr, cr = data.ingest_batch(samples, "cuda", channels=3)
        assert torch.equal(X, Xr) and torch.equal(Y, Yr) and torch.equal(F, Fr) and torch.equal(cls, cr)


# ---------------------------------------------------------------------------- f-2 ------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _meter_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    train.ddp_setup(backend="gloo")
    m = train.DeviceMeter(["loss", "acc"], device="cpu")
    for i in range(3 + rank):                                  # ranks see different numbers of batches
        m.update([torch.tensor(float(rank + i)), 0.1 * (i + 1)], weight=2 + rank)
    out[rank] = (m.averages(reduce=True), m.averages(reduce=False))
    dist.barrier()
    dist.destroy_process_group()


def test_device_meter_global_average_two_ranks():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_meter_worker, args=(world, port, out), nprocs=world, join=True)
    num_l = num_a = den = 0.0
    for rank in range(world):
        for i in range(3 + rank):
            w = 2 + rank
            num_l += w * (rank + i); num_a += w * 0.1 * (i + 1); den += w
    for rank in range(world):
        glob, local = out[rank]
        assert abs(glob["loss"] - num_l / den) < 1e-12 and abs(glob["acc"] - num_a / den) < 1e-12
        assert local != glob                                   # the per-rank view differs, the reduced one is common


# ---------------------------------------------------------------------------- f-4 ------------------------
def _toy_nets(seed):
    torch.manual_seed(seed)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4))   # noqa: E731
    return (mk(), mk(), None, mk(), mk())


def test_checkpoint_roundtrip_and_reference_file_names(tmp_path):
    nets = _toy_nets(0)
    opts = [train.FlatAdam(list(n.parameters()), lr=1e-3, lr_mult=1.0, zoom=False) for n in nets if n is not None]
    for k, o in enumerate(opts):
        o.t = 7 + k
        o.m.normal_(); o.v.uniform_()
    fovealseg.ops.DropoutState.step = 123
    train.save_checkpoint(str(tmp_path), 5, nets, opts, extra={"history": [1, 2, 3]})
    for name in ("encoder", "decoder", "saliency", "compress"):         # train_deform_semantic.py:166-184
        assert (tmp_path / f"{name}_epoch_5.pth").exists()
    want = [{k: v.clone() for k, v in n.state_dict().items()} for n in nets if n is not None]
    want_opt = [(o.t, o.m.clone(), o.v.clone()) for o in opts]

    nets2 = _toy_nets(1)
    opts2 = [train.FlatAdam(list(n.parameters()), lr=1e-3, lr_mult=1.0, zoom=False) for n in nets2 if n is not None]
    fovealseg.ops.DropoutState.step = 0
    extra = train.load_checkpoint(str(tmp_path), 5, nets2, opts2)
    assert extra == {"history": [1, 2, 3]} and fovealseg.ops.DropoutState.step == 123
    for n, sd in zip([n for n in nets2 if n is not None], want):
        for k, v in n.state_dict().items():
            assert torch.equal(v, sd[k]), k
    for o, (t, m, v) in zip(opts2, want_opt):
        assert o.t == t and torch.equal(o.m, m) and torch.equal(o.v, v)
    # parameters are still views into the optimiser arenas after loading
    for o in opts2:
        for p, off in zip(o.flat.params, o.flat.offsets):
            assert p.data_ptr() == o.flat.data.data_ptr() + 4 * off

    # a directory written by the reference holds the four state_dicts only
    os.remove(tmp_path / "train_state_epoch_5.pth")
    assert train.load_checkpoint(str(tmp_path), 5, _toy_nets(2)) is None
    assert train.history_path("d", "last", 3) == os.path.join("d", "history_epoch_last_3.csv")


# ---------------------------------------------------------------------------- f-3 ------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g14():
    return {k: v for k, v in np.load(os.path.join(GOLD, "g14_inverse.npz")).items()}


def _claims(grid, Hs, Ws):
    """per image: {pixel linear index: [grid point indices claiming it, ascending]}"""
    B, h, w, _ = grid.shape
    u = (((grid[..., 0] + 1) / 2) * (Ws - 1)).int().long().view(B, -1)
    v = (((grid[..., 1] + 1) / 2) * (Hs - 1)).int().long().view(B, -1)
    out = []
    for b in range(B):
        d = {}
        for i, p in enumerate((v[b] * Ws + u[b]).tolist()):
            d.setdefault(p, []).append(i)
        out.append(d)
    return out


def _check_inverse_grid(inv, ref, grid, Hs, Ws):
    """inv: ours (0 in holes), ref: the reference's grid_inv (NaN in holes).  Pixels claimed once must agree bit for bit.  For
    pixels claimed by several grid points the reference is not reproducible -- its two index_put_ calls (x and y channel,
    models/models.py:650-651) run on ATen's parallel CPU loop and may even pick different claimants per channel -- so there the
    reference value must come from SOME claimant and ours from the LAST one (the sequential index_put_ result)."""
    B, h, w, _ = grid.shape
    assert torch.equal(torch.isnan(ref[..., 0]), (inv == 0).all(-1) & torch.isnan(ref[..., 0])) and not torch.isnan(inv).any()
    claims = _claims(grid, Hs, Ws)
    fx = lambda i: torch.tensor(float(i % w)) / w * 2 - 1      # noqa: E731
    fy = lambda i: torch.tensor(float(i // w)) / h * 2 - 1     # noqa: E731
    ndup = 0
    for b in range(B):
        hole = torch.isnan(ref[b, ..., 0]).view(-1)
        assert set(torch.nonzero(~hole).view(-1).tolist()) == set(claims[b])
        for p, idx in claims[b].items():
            y, x = divmod(p, Ws)
            mine, theirs = inv[b, y, x], ref[b, y, x]
            assert float(mine[0]) == float(fx(idx[-1])) and float(mine[1]) == float(fy(idx[-1]))
            if len(idx) == 1:
                assert torch.equal(mine, theirs)
            else:
                ndup += 1
                assert float(theirs[0]) in {float(fx(i)) for i in idx} and float(theirs[1]) in {float(fy(i)) for i in idx}
    assert ndup > 100


def test_inverse_grid_oracle_vs_reference_golden():
    g = _g14()
    Hs, Ws = (int(v) for v in g["seg"])
    grid = torch.from_numpy(g["grid"])
    inv = O.inverse_grid_ref(grid, Hs, Ws)
    ref = torch.from_numpy(g["grid_inv"])
    assert torch.equal(torch.isnan(inv), torch.isnan(ref))
    _check_inverse_grid(torch.nan_to_num(inv), ref, grid, Hs, Ws)
    out, hole = O.unwarp_nearest_ref(torch.from_numpy(g["pred"]), torch.from_numpy(g["grid"]), Hs, Ws)
    assert torch.equal(hole, torch.from_numpy(g["unfilled"]))
    sampled = torch.from_numpy(g["sampled"])
    same = (torch.nan_to_num(inv) == torch.nan_to_num(ref)).all(-1) & ~hole      # claimed, and by the same grid point as in the reference
    keep = same[:, None].expand_as(out)
    assert torch.equal(out[keep], sampled[keep])                                # there: the reference's grid_sample values, bit for bit
    sampled = torch.where(keep, sampled, out)                                   # (elsewhere compare the fill against our own samples)
    # the fill is a nearest-neighbour fill whatever the tie rule: scipy's NearestNDInterpolator must see the same distances
    from scipy.interpolate import NearestNDInterpolator
    for b in range(out.shape[0]):
        pts = np.argwhere(~hole[b].numpy())
        q = np.argwhere(hole[b].numpy())
        sc = NearestNDInterpolator(pts, sampled[b, 0].numpy()[~hole[b].numpy()])(q)
        mine = out[b, 0].numpy()[hole[b].numpy()]
        d2 = O.nearest_distance_map(hole[b:b + 1])[0].numpy()[hole[b].numpy()]
        differ = sc != mine
        # where the two disagree, both picked a pixel at the same (minimal) distance: check mine is at distance d2
        vals = sampled[b, 0].numpy()
        for (y, x), m_, dd in zip(q[differ], mine[differ], d2[differ]):
            cand = [(yy, xx) for yy, xx in pts if (yy - y) ** 2 + (xx - x) ** 2 == dd]
            assert any(vals[yy, xx] == m_ for yy, xx in cand)
        assert differ.mean() < 0.5          # integer lattices have many equidistant pairs; each disagreement was checked above


@pytest.mark.gpu
def test_unwarp_nearest_matches_golden_and_oracle():
    g = _g14()
    Hs, Ws = (int(v) for v in g["seg"])
    grid, pred = torch.from_numpy(g["grid"]).cuda(), torch.from_numpy(g["pred"]).cuda()
    owner, inv = fovealseg.ops.inverse_grid(grid, Hs, Ws)
    ref_inv = torch.from_numpy(g["grid_inv"])
    assert torch.equal((owner < 0).cpu(), torch.isnan(ref_inv[..., 0]))
    _check_inverse_grid(inv.cpu(), ref_inv, torch.from_numpy(g["grid"]), Hs, Ws)
    assert torch.equal(inv.cpu(), torch.nan_to_num(O.inverse_grid_ref(torch.from_numpy(g["grid"]), Hs, Ws)))   # and = the sequential rule
    out, hole = fovealseg.ops.unwarp_nearest(pred, grid, Hs, Ws)
    want, whole = O.unwarp_nearest_ref(torch.from_numpy(g["pred"]), torch.from_numpy(g["grid"]), Hs, Ws)
    assert torch.equal(hole.cpu(), whole)
    assert torch.equal(out.cpu(), want)                                         # same samples, same nearest pixel, same tie rule


@pytest.mark.gpu
def test_fill_nearest_properties_full_size():
    # BASELINE-size frame: 1024 x 1024 from an 80 x 80 grid (99.4 % holes); size-independent properties
    torch.manual_seed(0)
    B, C, Hs, Ws = 2, 3, 1024, 1024
    grid = (torch.rand(B, 80, 80, 2) * 2 - 1).cuda()
    pred = torch.randn(B, C, 80, 80).cuda()
    out, hole = fovealseg.ops.unwarp_nearest(pred, grid, Hs, Ws)
    assert not torch.isnan(out).any() and float(hole.float().mean()) > 0.99
    out2 = out.clone()
    owner, _ = fovealseg.ops.inverse_grid(grid, Hs, Ws)
    scratch = torch.empty(2 * B * Hs * Ws, device="cuda", dtype=torch.int32)
    fovealseg.hip.call("fs_fill_nearest", out2.data_ptr(), owner.data_ptr(), scratch.data_ptr(), B, C, Hs, Ws)
    assert torch.equal(out, out2)                                               # idempotent
    vals = set(out[0, 0][~hole[0]].cpu().tolist())
    assert set(out[0, 0].unique().cpu().tolist()) <= vals                       # holes only ever take claimed values


@pytest.mark.gpu
@pytest.mark.parametrize("Hs,Ws,C", [(37, 300, 6), (50, 513, 3), (8, 1500, 1)])
def test_unwarp_nearest_ragged_widths(Hs, Ws, C):
    # widths that are no multiple of the row kernel's 256 segments, class counts around its 4-way unrolled copy
    g = torch.Generator().manual_seed(Hs * 1000 + Ws)
    grid = torch.rand(2, 9, 11, 2, generator=g) * 2.2 - 1.1
    grid = grid.clamp(-1, 1)
    pred = torch.randn(2, C, 9, 11, generator=g)
    out, hole = fovealseg.ops.unwarp_nearest(pred.cuda(), grid.cuda(), Hs, Ws)
    want, whole = O.unwarp_nearest_ref(pred, grid, Hs, Ws)
    assert torch.equal(hole.cpu(), whole)
    assert torch.equal(out.cpu(), want)


# ------------------------------------------------------------------------------------------------------------------
# f-1, the dataset contract: fovealseg.data.PreprocessDataset against records taken from the REFERENCE class on the same
# synthetic tree (tests/golden/make_dataset_pin.py -> g16_dataset.json / .npz)
# ------------------------------------------------------------------------------------------------------------------
def _dataset_pin():
    import importlib.util
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("_make_dataset_pin", os.path.join(here, "make_dataset_pin.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)               # defines TREE / write_tree; the reference is imported only by its main()
    with open(os.path.join(here, "g16_dataset.json")) as f:
        pin = json.load(f)
    assert [list(t) for t in mod.TREE] == pin["tree"]
    return mod, pin, np.load(os.path.join(here, "g16_dataset.npz"))


def _build_dataset(tmp_path):
    mod, pin, arrays = _dataset_pin()
    data_path, raw = mod.write_tree(str(tmp_path))
    ds = data.PreprocessDataset(data_path=data_path, marker="sp60000", dataset_partition="train", dataset_name="lvis", coco_root=raw)
    return ds, pin, arrays


def test_preprocess_dataset_file_contract_matches_reference(tmp_path):
    ds, pin, arrays = _build_dataset(tmp_path)
    assert (ds.HC, ds.WC, len(ds)) == (pin["HC"], pin["WC"], pin["len"])       # 640 / 640 / 5: the reference's 'cityscpaes' typo included
    mine = sorted(ds.data_info, key=lambda r: r["fpath_Y"])
    keys = ("idx_H", "idx_W", "Y_cls_s", "pad_left", "pad_right", "pad_top", "pad_bottom")
    for r, want in zip(mine, pin["data_info"]):
        assert os.path.relpath(r["fpath_Y"], str(tmp_path)) == want["fpath_Y"]
        assert os.path.relpath(r["fpath_X"], str(tmp_path)) == want["fpath_X"]   # incl. the val2017 / test2017 fallback lookups
        assert {k: r[k] for k in keys} == {k: want[k] for k in keys}
    # the CPU oracle of the ingest step (ingest_sample_ref) against the reference's own __getitem__ output for these samples
    for r, want in zip(mine, pin["data_info"]):
        s = ds[ds.data_info.index(r)]
        X, F2, Y, cls = O.ingest_sample_ref(s.img.numpy(), s.mask.numpy(), s.pads, s.focus, s.frame, s.cls)
        key = os.path.basename(r["fpath_Y"]).split(".")[0]
        assert list(X.shape) == want["X_shape"] and list(Y.shape) == want["Y_shape"]
        assert str(X.dtype) == want["X_dtype"] and str(Y.dtype) == want["Y_dtype"] and str(cls.dtype) == want["cls_dtype"]
        assert np.array_equal(X[:, ::37, ::41].numpy(), arrays[key + ":Xcrop"]) and np.array_equal(Y[:, ::37, ::41].numpy(), arrays[key + ":Ycrop"])
        assert float(X.double().sum()) == want["X_sum"] and float(Y.double().sum()) == want["Y_sum"]
        assert np.array_equal(F2.numpy(), arrays[key + ":F2"]) and int(cls[0]) == want["cls"]
    # batches() feeds DevicePrefetcher: lists of decoded samples, ragged tail kept
    sizes = [len(b) for b in ds.batches(2)]
    assert sizes == [2, 2, 1] and all(isinstance(s, data.Sample) for b in ds.batches(2) for s in b)


@pytest.mark.gpu
def test_preprocess_dataset_device_ingest_matches_reference(tmp_path):
    ds, pin, arrays = _build_dataset(tmp_path)
    mine = sorted(range(len(ds)), key=lambda i: ds.data_info[i]["fpath_Y"])
    samples = [ds[i] for i in mine]
    X, F2, Y, cls = data.ingest_batch(samples, device="cuda", channels=4)          # all five pad to 640 x 640: one batch
    assert X.shape == (5, 4, 640, 640) and Y.shape == (5, 1, 640, 640) and cls.dtype == torch.int64
    for b, want in enumerate(pin["data_info"]):
        key = os.path.basename(want["fpath_Y"]).split(".")[0]
        assert np.array_equal(X[b, :, ::37, ::41].cpu().numpy(), arrays[key + ":Xcrop"])      # bit-identical to the reference's floats
        assert np.array_equal(Y[b, :, ::37, ::41].cpu().numpy(), arrays[key + ":Ycrop"])
        assert float(X[b].double().sum()) == want["X_sum"] and float(Y[b].double().sum()) == want["Y_sum"]
        assert np.array_equal(F2[b].cpu().numpy(), arrays[key + ":F2"]) and int(cls[b, 0]) == want["cls"]
    # and through the prefetcher, as the training loop would consume it
    got = list(data.DevicePrefetcher(ds.batches(5, indices=mine), device="cuda", channels=4))
    assert len(got) == 1 and torch.equal(got[0][0], X) and torch.equal(got[0][2], Y)


@pytest.mark.gpu
def test_upsample_branch_full_resolution_accuracies():
    """MODEL.upsample=True inside forward (models/models.py:869-873,933-940,1074-1083): loss at the sampled resolution, the four
    accuracies at FULL resolution on the un-warped prediction (inverse grid + nearest fill), against the oracle's restatement."""
    cfg = fovealseg.lvis50_cfg()
    cfg.MODEL.upsample = True
    module, _ = train.build_module(cfg, device="cuda")
    module.eval()
    o = O.OracleDeformSeg()
    fovealseg.weights.apply_name_keyed_init(o)
    o.eval()
    X, Fp, Y, cls = train.synthetic_batch(2, 160, 160, seed=21, device="cpu")
    with torch.no_grad():
        got = module({"img_data": X.cuda(), "seg_label": Y.cuda(), "focus_point": Fp.cuda(), "cls_label": cls.cuda()}, is_inference=True)
        ref = o({"img_data": X, "seg_label": Y.clone(), "focus_point": Fp, "cls_label": cls}, is_inference=True, upsample=True)
        low = o({"img_data": X, "seg_label": Y.clone(), "focus_point": Fp, "cls_label": cls}, is_inference=True, upsample=False)
    got = np.array([float(v) for v in got])
    ref = np.array([float(v) for v in ref])
    assert len(got) == 6
    assert abs(got[0] - ref[0]) <= 2e-3 and abs(got[2] - ref[2]) <= 1e-5            # loss (unchanged by the branch), edge loss
    assert np.abs(got[[1, 3, 4, 5]] - ref[[1, 3, 4, 5]]).max() <= 5e-3, (got, ref)   # full-resolution accuracies
    assert abs(float(low[0]) - ref[0]) <= 1e-6                                        # the branch does not touch the loss
    with pytest.raises(NotImplementedError):
        bad = fovealseg.lvis50_cfg()
        bad.MODEL.upsample, bad.MODEL.rev_deform_interp = True, "tri"
        train.build_module(bad, device="cuda")


# ---------------------------------------------------------------------------- f-2 / f-4 on the device --------------
@pytest.mark.gpu
def test_device_meter_on_gpu_matches_running_average():
    """f-2: DeviceMeter fed with the step's 0-d CUDA outputs reproduces utils.AverageMeter's weighted running average
    (train_deform_semantic.py:100-123: update(val) per iteration, .average() at the end) without reading the device per step."""
    meter = train.DeviceMeter(["loss", "acc", "edge"], device="cuda")
    g = torch.Generator().manual_seed(5)
    vals = torch.rand(17, 3, generator=g, dtype=torch.float64)
    weights = [1.0 + (i % 3) for i in range(17)]
    for row, w in zip(vals, weights):
        meter.update([v.to("cuda") for v in row], weight=w)              # tensors stay on the device
    avg = meter.averages(reduce=False)
    want = (vals * torch.tensor(weights)[:, None]).sum(0) / sum(weights)
    for k, name in enumerate(("loss", "acc", "edge")):
        assert abs(avg[name] - float(want[k])) <= 1e-12
    meter.update([0.5, 0.25, torch.tensor(2.0)], weight=2.0)            # python floats / CPU tensors are accepted too
    assert abs(meter.averages(reduce=False)["loss"] - (float(want[0]) * sum(weights) + 1.0) / (sum(weights) + 2.0)) <= 1e-12


@pytest.mark.gpu
def test_checkpoint_of_the_real_module_on_gpu(tmp_path):
    """f-4 on the real module: the four files hold exactly the reference's state_dict keys (tests/golden/g15_state_dict.json, dumped
    from the reference module), a fresh module + optimisers resumed from them reproduce the saved one's next training step bit for bit
    in the deterministic part (forward loss) and carry the Adam state."""
    import json
    with open(os.path.join(GOLD, "g15_state_dict.json")) as f:
        pin = json.load(f)
    cfg = fovealseg.lvis50_cfg()
    module, nets = train.build_module(cfg, device="cuda")
    module.train()
    opts = train.create_optimizers(nets, cfg)
    try:
        batch = train.synthetic_batch(2, 256, 256, seed=8, device="cuda")
        fovealseg.ops.DropoutState.seed, fovealseg.ops.DropoutState.step = 9, 0
        train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=0)
        train.save_checkpoint(str(tmp_path), "last", nets, opts, extra={"iter": 1})
        want_keys = {"encoder": [], "decoder": [], "saliency": [], "compress": []}
        prefix = {"encoder.": "encoder", "decoder.": "decoder", "localization.": "saliency", "net_compress.": "compress"}
        for k, shape, _ in pin:
            for p, name in prefix.items():
                if k.startswith(p):
                    want_keys[name].append((k[len(p):], shape))
        for name, want in want_keys.items():
            sd = torch.load(tmp_path / f"{name}_epoch_last.pth", map_location="cpu", weights_only=True)
            assert [(k, list(v.shape)) for k, v in sd.items()] == want, name          # the reference's keys, order and shapes per file
        loss_next = float(train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=1)[0])

        module2, nets2 = train.build_module(cfg, device="cuda", init="random")
        module2.train()
        opts2 = train.create_optimizers(nets2, cfg)
        extra = train.load_checkpoint(str(tmp_path), "last", nets2, opts2)
        assert extra == {"iter": 1} and [o.t for o in opts2] == [1, 1, 1, 1]
        loss_resumed = float(train.train_step(module2, opts2, batch, cfg, epoch=1, cur_iter=1)[0])
        # same weights, same BN buffers, same dropout counter -> same forward; bwd-weight float atomics do not enter the forward
        assert abs(loss_resumed - loss_next) <= 1e-6 * max(1.0, abs(loss_next)), (loss_resumed, loss_next)
    finally:
        pass          # (round 1 reset a process-global here; the direct-gradient decision is per parameter now)


@pytest.mark.gpu
def test_nan_assertion_is_deferred_not_dropped():
    """models/models.py:721 asserts `not torch.isnan(xs).any()` mid-forward (a host sync per step).  Here the flag travels to pinned
    host memory without blocking and the same AssertionError surfaces at the end of train_step / eval_step, at the next forward, or
    on check_nan() -- never silently."""
    cfg = fovealseg.lvis50_cfg()
    module, nets = train.build_module(cfg, device="cuda")
    module.eval()
    good = train.synthetic_batch(2, 256, 256, seed=3, device="cuda")
    bad = tuple(t.clone() for t in good)
    bad[0][0, 0, 5, 7] = float("nan")                    # one NaN pixel in the image -> NaN saliency logits -> NaN softmax
    train.eval_step(module, good)                        # clean input: no assertion
    with pytest.raises(AssertionError, match="xs contains NaN values!"):
        train.eval_step(module, bad)
    train.eval_step(module, good)                        # the flag does not stick
    # a training step on the bad batch raises BEFORE the gradient exchange and the optimiser steps (the reference asserts mid-forward,
    # models/models.py:721): parameters, Adam moments and step counts are those of the previous step
    module.train()
    opts = train.create_optimizers(nets, cfg)
    train.train_step(module, opts, good, cfg, epoch=1, cur_iter=0)
    torch.cuda.synchronize()
    before = [(o.flat.data.clone(), o.m.clone(), o.v.clone(), o.t) for o in opts]
    with pytest.raises(AssertionError, match="xs contains NaN values!"):
        train.train_step(module, opts, bad, cfg, epoch=1, cur_iter=1)
    torch.cuda.synchronize()
    for o, (d, m, v, t) in zip(opts, before):
        assert torch.equal(o.flat.data, d) and torch.equal(o.m, m) and torch.equal(o.v, v) and o.t == t
        assert torch.isfinite(o.flat.data).all()
    train.train_step(module, opts, good, cfg, epoch=1, cur_iter=1)          # and the run continues
    assert all(bool(torch.isfinite(o.flat.data).all()) for o in opts)
    module.eval()
    # direct forward calls: the assertion of call k is raised by check_nan() or at the start of call k + 1
    X, Fp, Y, cls = bad
    feed = {"img_data": X[:, :3], "seg_label": Y, "focus_point": Fp, "cls_label": cls}
    with torch.no_grad():
        module(dict(feed), is_inference=True)
        with pytest.raises(AssertionError, match="xs contains NaN values!"):
            module.check_nan()
        module(dict(feed), is_inference=True)
        Xg, Fg, Yg, cg = good
        with pytest.raises(AssertionError, match="xs contains NaN values!"):
            module({"img_data": Xg[:, :3], "seg_label": Yg, "focus_point": Fg, "cls_label": cg}, is_inference=True)


@pytest.mark.gpu
def test_training_trajectory_same_in_every_conv_mode():
    """40 optimisation steps on one fixed batch (train mode, dropout replayed by the step counter): the loss falls by more than a third
    in every conv mode, and the split-precision trajectories (F(2,3) row kernel included) stay as close to the fp32-MFMA mode's as a
    second run of the fp32-MFMA mode itself does.  The bound is derived from that noise, measured inside the test: the f32
    mode runs TWICE (bwd-weight float atomics change the last bits of a gradient, Adam's g / sqrt(v) turns that into +-lr on near-zero
    gradients: 1.3e-2 after 20 steps in tools/trajectory_probe.py), and every other mode must stay within 3 x max(that spread, 1.5e-2)
    of the first f32 curve at every step.  A wrong gradient anywhere in the conv engine (a wrong scale on one tap class, say) separates
    the curves by far more, or stops the descent."""
    cfg = fovealseg.lvis50_cfg()
    curves = {}
    try:
        for name, mode in (("f32", "f32"), ("f32_again", "f32"), ("bf16x3", "bf16x3"), ("f16x2", "f16x2")):
            fovealseg.hip.set_conv_precision(mode)
            module, nets = train.build_module(cfg, device="cuda")
            module.train()
            opts = train.create_optimizers(nets, cfg)
            batch = train.synthetic_batch(4, 256, 256, seed=11, device="cuda")
            fovealseg.ops.DropoutState.seed, fovealseg.ops.DropoutState.step = 5, 0
            losses = []
            for it in range(40):
                out = train.train_step(module, opts, batch, cfg, epoch=1, cur_iter=it)
                losses.append(out[0].detach().reshape(-1)[0])
            curves[name] = torch.stack(losses).double().cpu()
            del module, nets, opts
    finally:
        fovealseg.hip.set_conv_precision(fovealseg.hip.default_conv_precision())
    ref = curves["f32"]
    spread = float(((curves["f32_again"] - ref).abs() / ref.abs()).max())
    # Training a 130 M-parameter net on ONE batch of four images is chaotic: the last-bit noise of the bwd-weight atomics grows from step
    # to step (two f32 runs of the same 40 steps: spread 0.01-0.03 in most runs, 0.10 seen once; in deterministic mode they are bit-identical,
    # tests/test_ddp_gloo.py).  So the check has two parts: (a) over the FIRST TEN steps, before the noise has grown, every mode stays
    # within a fixed 2e-2 of the f32 curve (measured 1.4e-3 / 1.9e-3 / 4.4e-3) -- a wrong gradient anywhere separates the curves from step 2 on; (b) over all 40 steps within
    # 3 x the measured spread, capped (the bound must not widen without limit, and the f32 noise itself has a ceiling).
    assert spread <= 0.15, ("two f32 runs of the same 40 steps drifted apart by more than atomics-order noise explains", spread)
    early = float(((curves["f32_again"] - ref).abs() / ref.abs())[:10].max())
    assert early <= 2e-2, ("f32 twice, first ten steps", early)
    bound = min(3.0 * max(spread, 1.5e-2), 0.2)
    for name, c in curves.items():
        assert torch.isfinite(c).all() and float(c[-3:].mean()) < (2.0 / 3.0) * float(c[:3].mean()), (name, c)
        assert abs(float(c[0]) - float(ref[0])) <= 1e-4 * float(ref[0]), (name, float(c[0]), float(ref[0]))      # same forward before any update
        rel = (c - ref).abs() / ref.abs()
        print(f"trajectory {name}: first ten steps {float(rel[:10].max()):.2e}, all {float(rel.max()):.2e} (f32 spread {spread:.2e}, bound {bound:.2e})")
        assert float(rel[:10].max()) <= 2e-2, (name, "first ten steps", float(rel[:10].max()))
        assert float(rel.max()) <= bound, (name, float(rel.max()), spread, bound, c, ref)
