"""Training-step plumbing around DeformSegmentationModule.

Mirrors train_deform_semantic.py: `ddp_setup` (:45-55), the per-step order of `train()` (:74-129:
zero_grad -> adjust_learning_rate -> forward -> loss.mean().backward() -> 4 optimiser steps),
`create_optimizers` (:260-290, Adam x4 over encoder/decoder/saliency/compress) and
`adjust_learning_rate` (:302-350).

MI355X-first differences (DESIGN.md "Multi-GPU"): each optimiser owns ONE flat fp32 arena holding all
its parameters (params are views into it, conv weights keep RSCK strides), one flat gradient arena
and flat Adam moments, so a step is one fused HIP kernel and the data-parallel exchange is one RCCL
all-reduce per arena (4 per step, 522 MB total) instead of DDP's ~20 x 25 MB buckets.
"""
import os

import torch
import torch.distributed as dist

from . import hip

_ALIGN = 4   # elements; keeps every parameter 16-byte aligned for the float4 kernels


class FlatParams:
    """Re-homes a list of parameters into one flat arena (+ a matching gradient arena)."""

    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, "empty parameter list"
        dev = self.params[0].device
        self._epoch = [0]          # bumped whenever the arena is rewritten behind the parameters' version counters (refresh_amax)
        offs, n = [], 0
        for p in self.params:
            assert p.dtype == torch.float32
            offs.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.data = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, offs):
            size, stride = tuple(p.shape), tuple(p.stride())
            dense = self._dense_strides(p)
            view = self.data.as_strided(size, dense, o)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad.as_strided(size, dense, o)
            p._fs_grad_home = (self.grad, o)          # ops._direct_grad_target: kernels may add into this slice while .grad points at it
            p._fs_epoch = self._epoch                  # ops._pack_for: a weight pack of p is valid while this word has not moved
            del stride
        self.offsets = offs
        # max|w| bits per parameter, kept current by refresh_amax(): the f16x2 conv kernels scale the weights by it
        self.amax = torch.zeros(len(self.params), device=dev, dtype=torch.int32)
        self._offs_dev = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._sizes_dev = torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev)
        self._amax_words = [self.amax[i:i + 1] for i in range(len(self.params))]
        self.refresh_amax()

    def refresh_amax(self):
        """One launch: max|w| of every parameter of the arena.  Parameters remember their word and the torch version counter
        at this moment (ops.weight_amax drops the word if the tensor is modified through torch afterwards)."""
        self._epoch[0] += 1
        if not self.data.is_cuda:
            return
        hip.call("fs_weight_amax_segments", hip.ptr(self.data), hip.ptr(self._offs_dev), hip.ptr(self._sizes_dev), len(self.params),
                 hip.ptr(self.amax))
        for i, p in enumerate(self.params):
            p._fs_amax = (self._amax_words[i], p._version)

    @staticmethod
    def _dense_strides(p):
        """Strides of p if it is non-overlapping and dense (any permutation), else contiguous ones."""
        size, stride = p.shape, p.stride()
        order = sorted(range(p.dim()), key=lambda d: (stride[d], size[d]))
        expect, ok = 1, True
        for d in order:
            if size[d] == 1:
                continue
            if stride[d] != expect:
                ok = False
                break
            expect *= size[d]
        if ok:
            return tuple(stride)
        return tuple(torch.empty(size).stride())

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):     # autograd may have re-pointed .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad.as_strided(tuple(p.shape), tuple(p.stride()), o)

    def adopt_grads(self):
        """Copy gradients that live outside the arena (a caller followed the reference loop's `module.zero_grad()`, so autograd
        allocated fresh `.grad` tensors) into the arena and re-point `.grad` at the views."""
        for p, o in zip(self.params, self.offsets):
            view = self.grad.as_strided(tuple(p.shape), tuple(p.stride()), o)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                view.copy_(p.grad)
            p.grad = view


class FlatAdam:
    """torch.optim.Adam(weight_decay) semantics over a FlatParams arena, one fused HIP launch."""

    def __init__(self, params, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, **group_extras):
        self.flat = FlatParams(params)
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self.t = 0
        self.grad_scale = 1.0
        g = dict(params=self.flat.params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        g.update(group_extras)
        self.param_groups = [g]

    def zero_grad(self, set_to_none=False):
        from . import ops
        ops.join_wgrad_streams()          # (a backward whose small-layer weight gradients went to the side stream, never stepped)
        self.flat.zero_grad()

    def check_grads_in_arena(self):
        """Every parameter's .grad must BE its slice of the gradient arena: the fused Adam kernel reads the arena only.  A caller
        that dropped the views (module.zero_grad() sets .grad = None, autograd then allocates fresh tensors) would otherwise
        step with zero gradients and weight decay alone, silently."""
        base = self.flat.grad.data_ptr()
        for p, o in zip(self.flat.params, self.flat.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                raise RuntimeError("FlatAdam: a parameter's .grad is not its gradient-arena view (use optimizer.zero_grad(), not "
                                   "module.zero_grad(); or call flat.adopt_grads() to copy stray gradients in)")

    def step(self):
        g = self.param_groups[0]
        self.check_grads_in_arena()
        self.t += 1
        from .ops import _launch, join_wgrad_streams
        join_wgrad_streams()              # the arena is complete only when the side-stream weight gradients have landed
        _launch("fe_adam", 28.0 * self.flat.numel, "fs_adam_step", hip.ptr(self.flat.data), hip.ptr(self.flat.grad), hip.ptr(self.m), hip.ptr(self.v),
                self.flat.numel, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                float(g["weight_decay"]), self.t, float(self.grad_scale))
        self.flat.refresh_amax()

    def state_dict(self):
        return dict(t=self.t, m=self.m, v=self.v, param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])

    def load_state_dict(self, sd):
        self.t = sd["t"]
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])


def create_optimizers(nets, cfg):
    """train_deform_semantic.py:260-290 -- returns (encoder, decoder, saliency, compress) optimisers."""
    net_encoder, net_decoder, crit, net_saliency, net_compress = nets
    if cfg.TRAIN.optim.lower() != "adam":
        raise NotImplementedError("only TRAIN.optim='adam' is usable in the reference (the sgd branch returns undefined names)")
    T = cfg.TRAIN
    mk = lambda net, mult, zoom: FlatAdam(list(net.parameters()), lr=T.lr_encoder, weight_decay=T.weight_decay,  # noqa: E731
                                          lr_mult=mult, zoom=zoom)
    return (mk(net_encoder, T.lr_mult_encoder, False), mk(net_decoder, T.lr_mult_decoder, False),
            mk(net_saliency, T.lr_mult_saliency, True), mk(net_compress, T.lr_mult_compress, True))


def adjust_learning_rate(optimizers, cur_iter, cfg, epoch=None):
    """train_deform_semantic.py:302-350 under scale_by_iter=False, fov_scale_lr=''."""
    T = cfg.TRAIN
    scale_running_lr = (1.0 - float(cur_iter) / T.max_iters) ** T.lr_pow
    T.running_lr_encoder = T.lr_encoder * scale_running_lr
    T.running_lr_decoder = T.lr_decoder * scale_running_lr
    T.running_lr_foveater = T.lr_foveater * scale_running_lr
    base_lr = 0.1
    n_pre = T.deform_pretrain
    if T.scale_by_iter:
        raise NotImplementedError("TRAIN.scale_by_iter=True is not on the default path")
    lr_idx = epoch
    if T.deform_pretrain_bol or lr_idx < n_pre:
        lr_class = base_lr * 0.1 ** (lr_idx // n_pre)
        lr_zoom = base_lr * 0.1 ** (lr_idx // n_pre)
    else:
        lr_class = base_lr * 0.1 ** ((lr_idx - n_pre) // n_pre)
        lr_zoom = base_lr * 0.1 ** (lr_idx // n_pre)
    if T.fix_deform_aft_pretrain and T.fix_deform_start_epoch <= epoch <= T.fix_deform_end_epoch:
        lr_zoom = 0.0
    for opt in optimizers:
        for group in opt.param_groups:
            group["lr"] = group["lr_mult"] * (lr_zoom if group["zoom"] else lr_class)


# ----------------------------------------------------------------------------------------------
# data-parallel (one process per GPU, RCCL over xGMI)
# ----------------------------------------------------------------------------------------------
def ddp_setup(backend=None):
    """train_deform_semantic.py:45-55, driven by torchrun's env instead of mp.spawn arguments."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        kw = {}
        if backend == "nccl" and torch.cuda.is_available():      # bind the communicator (and barrier()) to this rank's GPU
            kw["device_id"] = torch.device("cuda", local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    return rank, local_rank, world


def _collectives_on():
    """A process group is up.  A group of ONE rank still runs its collectives (they are cheap no-op exchanges): the same
    code path -- device binding, RCCL's stream hand-off behind the side-stream backward -- is then exercised on a single GPU."""
    return dist.is_available() and dist.is_initialized()


def _is_torch_ddp(module):
    return module is not None and isinstance(module, torch.nn.parallel.DistributedDataParallel)


def broadcast_parameters(optimizers, module=None, src=0):
    """DDP construction-time broadcast (train_deform_semantic.py:395): one collective per arena."""
    if not _collectives_on():
        return
    for opt in optimizers:
        dist.broadcast(opt.flat.data, src=src)
        opt.flat.refresh_amax()          # the arena was rewritten behind the parameters' version counters
    if module is not None:
        broadcast_buffers(module, src=src)
    # (No device-wide synchronise here.  Round 4 added one on the guess that a gloo broadcast of the last two arenas "had not landed" in one
    # red run of the 2-rank wrapper test; the record of that run refutes it -- the encoder / decoder arenas agreed to 2e-7, so both passes
    # ran the same parameters and the same forward -- and a synchronous collective already orders the CURRENT stream behind the backend's
    # copy-back (gloo: events on its copy streams; RCCL: its own stream), which is the stream refresh_amax and the next forward run on,
    # and every side stream of this package forks from that stream by an event.  DESIGN.md 6 has the launch-by-launch ordering table.)


class FlatBuffers:
    """Every float32 buffer of `module` (BatchNorm running statistics, the reference SyncBN's three extra buffers, the Gaussian
    filter) re-homed into ONE flat tensor; the registered buffers become views, so the kernels that update running statistics
    through their pointers keep working and the per-step buffer broadcast is a single collective."""

    def __init__(self, module):
        bufs = [b for _, b in module.named_buffers() if b.dtype == torch.float32]
        assert bufs, "module has no float buffers"
        offs, n = [], 0
        for b in bufs:
            offs.append(n)
            n += (b.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.data = torch.zeros(n, device=bufs[0].device, dtype=torch.float32)
        for b, o in zip(bufs, offs):
            view = self.data[o:o + b.numel()].view(b.shape)
            view.copy_(b)
            b.data = view
        self.buffers, self.offsets = bufs, offs


def broadcast_buffers(module, src=0):
    """DDP(broadcast_buffers=True) semantics (train_deform_semantic.py:395; torch DDP syncs module buffers from rank 0 at the start of
    every training forward): after this call every rank holds rank 0's BatchNorm running statistics.  One collective over the
    module's flat buffer arena (built on first use)."""
    if not _collectives_on():
        return
    flat = getattr(module, "_fs_flat_buffers", None)
    if flat is None:
        flat = FlatBuffers(module)
        object.__setattr__(module, "_fs_flat_buffers", flat)      # not a submodule / buffer: keep it out of state_dict
    dist.broadcast(flat.data, src=src)


COMM_EVENTS = None      # bench.py: a list that receives one (start, stop) HIP-event pair per gradient exchange (the four arena all-reduces)


def allreduce_gradients(optimizers, module=None):
    """Gradient average across ranks: one SUM all-reduce per arena; the 1/world is folded into Adam.
    `module` wrapped by torch's DistributedDataParallel (the reference's call site, train_deform_semantic.py:395): its reducer has
    already averaged the gradients into the arena views during backward, so nothing is exchanged here."""
    from . import ops
    ops.join_wgrad_streams()          # weight gradients of small layers may still be running on their side stream
    if not _collectives_on() or _is_torch_ddp(module):
        for opt in optimizers:
            opt.grad_scale = 1.0
        return
    world = dist.get_world_size()
    timed = COMM_EVENTS is not None and optimizers[0].flat.grad.is_cuda
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    for opt in optimizers:
        dist.all_reduce(opt.flat.grad, op=dist.ReduceOp.SUM)
        opt.grad_scale = 1.0 / world
    if timed:
        ev1.record()          # (the process group's stream is joined to the current stream by the collective's own event hand-off)
        COMM_EVENTS.append((ev0, ev1))


def shard_indices(n_samples, rank, world, epoch_seed=0, shuffle=True):
    """DistributedSampler(num_replicas, rank, shuffle=True) sharding (train_deform_semantic.py:462);
    set_epoch is never called in the reference, so the permutation is the same every epoch (seed 0)."""
    g = torch.Generator().manual_seed(epoch_seed)
    idx = torch.randperm(n_samples, generator=g).tolist() if shuffle else list(range(n_samples))
    total = (n_samples + world - 1) // world * world
    idx += idx[: total - len(idx)]
    return idx[rank:total:world]


# ----------------------------------------------------------------------------------------------
# one optimisation step / one evaluation step
# ----------------------------------------------------------------------------------------------
def train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=0):
    """train_deform_semantic.py:74-129 for one batch (X,F,Y,cls) already on the device."""
    from .ops import DropoutState
    X, Fp, Y, cls = batch
    feed = {"img_data": X[:, :3], "seg_label": Y, "focus_point": Fp, "cls_label": cls}
    for opt in optimizers:
        opt.zero_grad()
    adjust_learning_rate(optimizers, cur_iter, cfg, epoch=epoch)
    DropoutState.step += 1
    ddp = _is_torch_ddp(module)
    inner = module.module if ddp else module
    if not ddp:
        broadcast_buffers(module)      # no-op without a process group; torch DDP syncs buffers from rank 0 itself before every forward
    out = module(feed, epoch=epoch, cur_iter=cur_iter)
    loss = out[0]
    loss.mean().backward()
    # models/models.py:721 asserts on NaN saliency in the middle of the forward, i.e. BEFORE any update; the deferred flag is read
    # here, after backward (the saliency forward finished long ago: no stall) and before the gradient exchange and the optimiser
    # steps, so a caller that catches the AssertionError still holds the parameters and Adam moments of the previous step
    if hasattr(inner, "check_nan"):
        inner.check_nan()
    allreduce_gradients(optimizers, module)
    T = cfg.TRAIN
    for opt in optimizers:          # train_deform_semantic.py:112-120: which optimisers step in this epoch
        zoom = opt.param_groups[0]["zoom"]
        if T.fix_deform_aft_pretrain and T.fix_deform_start_epoch <= epoch <= T.fix_deform_end_epoch:
            if zoom:                # deformation module frozen: its Adam state (t, m, v) must not advance either
                continue
        elif T.opt_deform_LabelEdge and T.fix_seg_start_epoch <= epoch <= T.fix_seg_end_epoch:
            if not zoom:            # segmentation module frozen
                continue
        opt.step()
    from . import ops
    ops.repack_weights()      # the conv layers' weight packs of the next step, on a side stream beside its front end
    return out


@torch.no_grad()
def eval_step(module, batch):
    """eval.py:389-405 -- same forward with is_inference=True under no_grad (module.eval() by caller)."""
    X, Fp, Y, cls = batch
    feed = {"img_data": X[:, :3], "seg_label": Y, "focus_point": Fp, "cls_label": cls}
    out = module(feed, is_inference=True)
    if hasattr(module, "check_nan"):
        module.check_nan()
    return out


def synthetic_batch(B, H, W, seed=1, device="cuda"):
    """uniform RGB, gaze in [0.1,0.9), disc mask of radius 0.15 H at the gaze."""
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(B, 3, H, W, generator=g)
    Fp = torch.rand(B, 2, generator=g) * 0.8 + 0.1
    cls = torch.randint(0, 50, (B, 1), generator=g)
    ii = torch.arange(H, dtype=torch.float32)[None, :, None]
    jj = torch.arange(W, dtype=torch.float32)[None, None, :]
    cy = (Fp[:, 0] * (H - 1))[:, None, None]
    cx = (Fp[:, 1] * (W - 1))[:, None, None]
    Y = (((ii - cy) ** 2 + (jj - cx) ** 2) <= (0.15 * H) ** 2).float().unsqueeze(1)
    return X.to(device), Fp.to(device), Y.to(device), cls.to(device)


def build_module(cfg, device="cuda", init="name_keyed"):
    from . import ModelBuilder, DeformSegmentationModule
    from .weights import apply_name_keyed_init
    M = cfg.MODEL
    enc = ModelBuilder.build_encoder(M.arch_encoder, M.fc_dim, M.weights_encoder)
    dec = ModelBuilder.build_decoder(M.arch_decoder, M.fc_dim, cfg.DATASET.num_class, M.weights_decoder)
    sal = ModelBuilder.build_net_saliency(cfg, M.weights_net_saliency)
    comp = ModelBuilder.build_net_compress(cfg, M.weights_net_compress)
    module = DeformSegmentationModule(enc, dec, sal, comp, None, cfg)
    if init == "name_keyed":
        apply_name_keyed_init(module)
    module.to(device)
    nets = (module.encoder, module.decoder, None, module.localization, module.net_compress)
    return module, nets


# ----------------------------------------------------------------------------------------------
# metrics step after the path
# ----------------------------------------------------------------------------------------------
class DeviceMeter:
    """Running averages of the step outputs (loss, acc, edge loss, ...) kept ON the device: `update` is an in-place add of
    0-d tensors (no `.item()` sync per iteration, unlike utils.AverageMeter at train_deform_semantic.py:100-123), and
    `averages(reduce=True)` makes the one host read, after ONE all-reduce of [sums..., count] over the ranks -- the
    reference keeps its meters per rank and never reduces them."""

    def __init__(self, names, device):
        self.names = list(names)
        self.acc = torch.zeros(len(self.names) + 1, device=device, dtype=torch.float64)
        self._sums, self._count = self.acc[:-1], self.acc[-1:]       # views, made once

    def update(self, values, weight=1.0):
        dev = self.acc.device
        if all(torch.is_tensor(x) and x.device == dev and x.dim() == 0 for x in values):
            v = torch.stack([x.detach() for x in values])          # the step's outputs: one stack + two in-place adds per step
        else:
            v = torch.stack([(x.detach().to(device=dev, dtype=torch.float64) if torch.is_tensor(x) else torch.tensor(float(x), dtype=torch.float64, device=dev)).reshape(())
                             for x in values])
        self._sums.add_(v, alpha=weight)
        self._count.add_(weight)

    def averages(self, reduce=True):
        tot = self.acc.clone()
        if reduce and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot = tot.cpu()
        n = float(tot[-1])
        return {k: (float(tot[i]) / n if n > 0 else float("nan")) for i, k in enumerate(self.names)}


# ----------------------------------------------------------------------------------------------
# checkpoint / resume
# ----------------------------------------------------------------------------------------------
_NET_FILES = ("encoder", "decoder", "saliency", "compress")


def save_checkpoint(dirpath, epoch, nets, optimizers=None, extra=None):
    """train_deform_semantic.py:166-184 -- `{encoder,decoder,saliency,compress}_epoch_{N}.pth` hold plain state_dicts with the
    reference's keys and shapes (they load into the reference and vice versa).  In addition (the reference saves neither, so a
    resumed run restarts Adam from zero) `train_state_epoch_{N}.pth` keeps the four Adam states, the dropout step counter and
    the torch RNG state.  epoch may be 'last' (checkpoint_last, :188-208)."""
    from .ops import DropoutState
    net_encoder, net_decoder, _crit, net_saliency, net_compress = nets
    os.makedirs(dirpath, exist_ok=True)
    for name, net in zip(_NET_FILES, (net_encoder, net_decoder, net_saliency, net_compress)):
        torch.save({k: v.detach().cpu().contiguous() for k, v in net.state_dict().items()}, os.path.join(dirpath, f"{name}_epoch_{epoch}.pth"))
    state = {"epoch": epoch, "dropout": {"seed": DropoutState.seed, "step": DropoutState.step}, "rng": torch.get_rng_state(),
             "extra": extra}
    if optimizers is not None:
        state["optimizers"] = [{"t": o.t, "m": o.m.cpu(), "v": o.v.cpu(), "lr": o.param_groups[0]["lr"]} for o in optimizers]
    torch.save(state, os.path.join(dirpath, f"train_state_epoch_{epoch}.pth"))


def load_checkpoint(dirpath, epoch, nets, optimizers=None, strict=True):
    """Inverse of save_checkpoint; also accepts a directory written by the reference (no train_state file: weights only).
    Returns the `extra` object (or None)."""
    from .ops import DropoutState
    net_encoder, net_decoder, _crit, net_saliency, net_compress = nets
    for name, net in zip(_NET_FILES, (net_encoder, net_decoder, net_saliency, net_compress)):
        sd = torch.load(os.path.join(dirpath, f"{name}_epoch_{epoch}.pth"), map_location="cpu", weights_only=True)
        net.load_state_dict(sd, strict=strict)
    path = os.path.join(dirpath, f"train_state_epoch_{epoch}.pth")
    if not os.path.exists(path):
        if optimizers is not None:
            for o in optimizers:
                o.flat.refresh_amax()
        return None
    state = torch.load(path, map_location="cpu", weights_only=True)      # plain tensors / numbers / strings only (see save_checkpoint)
    DropoutState.seed, DropoutState.step = state["dropout"]["seed"], state["dropout"]["step"]
    torch.set_rng_state(state["rng"])
    if optimizers is not None:
        for o, sd in zip(optimizers, state.get("optimizers", [])):
            o.load_state_dict(sd)
            o.param_groups[0]["lr"] = sd["lr"]
            o.flat.refresh_amax()          # the parameters were rewritten through load_state_dict
    return state.get("extra")


def history_path(dirpath, epoch, rank=0, kind="csv"):
    """The reference writes `history_epoch_last_{rank}.csv` / `history_epoch_{N}_{rank}.pth` (train…:230-235) but resumes from
    `history_epoch_{N}.csv` (:416), which never exists.  Writers and readers here share this one function."""
    return os.path.join(dirpath, f"history_epoch_{epoch}_{rank}.{kind}")
