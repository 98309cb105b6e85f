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
