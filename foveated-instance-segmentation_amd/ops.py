"""torch.autograd bindings of the HIP kernels (one Function per kernel group).

Activations travel as contiguous NHWC fp32 tensors (B,H,W,C).  Convolution weights keep the
reference's logical (Cout,Cin,R,S) shape with RSCK strides, so `w.permute(2,3,1,0)` is the contiguous
[R][S][Cin][Cout] array the kernels read (DESIGN.md "Data layout").
"""
import ctypes
import os
import weakref
import zlib

import torch
from torch.autograd import Function

from . import hip

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
BN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# dropout keys (integer hash shared with csrc/common.h and the oracle replay)
# ----------------------------------------------------------------------------------------------
class DropoutState:
    """Global (seed, step) pair; every conv layer derives its key from it and its own id."""
    seed = 0
    step = 0

    @classmethod
    def key(cls, layer_id: int) -> int:
        return layer_key(cls.seed * 1000003 + cls.step, layer_id)


def layer_key(seed: int, layer_id: int) -> int:
    x = (seed * 0x9E3779B1 + layer_id * 0x85EBCA6B + 0x27D4EB2F) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    x = (x * 0x297A2D39) & 0xFFFFFFFF
    x ^= x >> 15
    return x


def layer_id_from_name(name: str) -> int:
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


# ----------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------
def rsck(w: torch.Tensor) -> torch.Tensor:
    """Contiguous [R][S][Cin][Cout] view of a logical (Cout,Cin,R,S) weight."""
    v = w.permute(2, 3, 1, 0)
    if not v.is_contiguous():
        raise hip.HipLibraryError(f"conv weight {tuple(w.shape)} is not stored RSCK (strides {w.stride()})")
    return v


def new_rsck_weight(cout, cin, r, s, device=None) -> torch.Tensor:
    """Uninitialised logical (Cout,Cin,R,S) tensor with RSCK storage."""
    return torch.empty(r, s, cin, cout, device=device, dtype=torch.float32).permute(3, 2, 0, 1)


def _out_hw(h, w, r, s, stride, pad, dil=1):
    return (h + 2 * pad - dil * (r - 1) - 1) // stride + 1, (w + 2 * pad - dil * (s - 1) - 1) // stride + 1


class KernelTimer:
    """Optional HIP-event timing of individual launches on the current stream (bench.py only).
    `flops` is the algorithmic work of the launch: 2 * B*Ho*Wo * Cout * R*S*Cin for every conv variant, bytes for the
    HBM-bound kinds (bn_fwd, bn_bwd)."""

    def __init__(self):
        self.records = {}          # kind -> list of (start_event, stop_event, flops)
        self.tags = {}             # kind -> list of (entry point, leading integer arguments) per record

    def launch(self, kind, flops, fn, tag=None):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.setdefault(kind, []).append((s, e, flops))
        self.tags.setdefault(kind, []).append(tag)

    def by_shape(self):
        """{(entry point, int args...): [launches, total_ms, flops]} over all recorded launches."""
        out = {}
        for kind, recs in self.records.items():
            for (s, e, f), tag in zip(recs, self.tags[kind]):
                d = out.setdefault(tag, [0, 0.0, 0.0])
                d[0] += 1; d[1] += s.elapsed_time(e); d[2] += f
        return out

    def summary(self):
        out = {}
        for kind, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[kind] = dict(launches=len(recs), total_ms=ms, flops=fl)
        return out


WGRAD_FIRST = os.environ.get("FS_WGRAD_FIRST", "1") != "0"      # kernel experiments: order of the two conv backward launches
TIMER = None      # set to a KernelTimer by bench.py around the timed region
# Parity-test instrument: when a list, every activation site appends (BatchNorm module | (HighResolutionModule, i), act kind,
# activated output), so a test can replay the branch each activation took in the CPU oracle (oracle ACT_REPLAY).
ACT_TRACE = None
# Flight recorder of the front-end backward (tests / probes): when a dict, DeformSegmentationModule.forward hooks the cotangents that
# reach x_sampled, the sampling grid and xs -- the grid path's and the edge loss's contributions to xs separately, and their sum -- and
# every hook leaves {name: (fp64 norm, int64 sum of the bit patterns)} there as DEVICE scalars (no host read inside backward).  Every
# kernel on that chain is order-fixed, so two passes over the same inputs must agree bit for bit, whichever way the weight gradients
# travel (arena-direct + side stream, or AccumulateGrad under torch DDP): a mismatch names the stage (tests/test_ddp_gloo.py).
GRAD_TRACE = None


def grad_probe(t, name, alias=False):
    """Record the gradient arriving at `t` under `name` in GRAD_TRACE; alias=True returns a view of t so that ONE consumer's
    contribution is seen on its own."""
    if GRAD_TRACE is None or not t.requires_grad:
        return t
    trace = GRAD_TRACE
    if alias:
        t = t.view_as(t)

    def hook(g):
        gd = g.detach()
        trace[name] = (gd.double().norm(), gd.contiguous().view(torch.int32).to(torch.int64).sum())
    t.register_hook(hook)
    return t


def _launch(kind, flops, name, *args):
    if TIMER is None:
        hip.call(name, *args)
    else:
        TIMER.launch(kind, flops, lambda: hip.call(name, *args), tag=(name,) + tuple(a for a in args if type(a) is int and a < (1 << 20)))


def _conv_kind(Cs, Cd, R, S, stride, pad, dil):
    """Timer bucket = the kernel csrc/conv.hip selects: the halo-tiled 3x3 stride-1 kernel, the other aligned kernels, or the generic one."""
    if Cs % 4 or Cd % 4:
        return "conv_generic"
    if R == 3 and S == 3 and stride == 1 and pad == 1 and dil == 1 and Cs >= 32 and hip.get_conv_precision() != "f32":
        return "conv3x3"
    return "conv_affine"


def weight_amax(w):
    """Device word holding max|w| bits of parameter w if its optimiser keeps one and w has not been modified since
    (train.FlatParams.refresh_amax), else None: the library then reduces over the weights itself."""
    rec = getattr(w, "_fs_amax", None)
    if rec is not None and rec[1] == w._version:
        return rec[0]
    if torch.is_grad_enabled() or not w.is_cuda or w.dim() != 4 or hip.get_conv_precision() != "f16x2":
        return None
    # inference without an optimiser arena: reduce once per weight version instead of once per call (272 memset + reduction
    # launches per HRNet forward otherwise)
    word = torch.empty(1, dtype=torch.int32, device=w.device)
    offs = torch.zeros(1, dtype=torch.int64, device=w.device)
    sizes = torch.full((1,), w.numel(), dtype=torch.int64, device=w.device)
    hip.call("fs_weight_amax_segments", hip.ptr(rsck(w)), hip.ptr(offs), hip.ptr(sizes), 1, hip.ptr(word))
    w._fs_amax = (word, w._version)
    return word


# The conv kernels address a tensor with 32-bit byte offsets: an activation or gradient tensor of 4 GB or more (configs[3] at B = 64: the C1
# head's 1024-channel input at 160 x 160 is 6.7 GB) is processed in batch ranges that fit -- the images of a batch are independent in all
# three convolutions, BatchNorm partial-sum slabs simply concatenate, weight gradients accumulate.  Only the fused extras of the bwd-data
# epilogue (BatchNorm sums, residual addend) are given up on that path.  Dropout masks hash the element index of the FULL tensor, which
# the library cannot reproduce from a sub-range: such calls are not split (and fail loudly in the library, as before).
MAX_TENSOR_BYTES = 4294967000


def _batch_ranges(B, *per_image_elems):
    """[(b0, b1), ...] with every listed per-image element count x (b1 - b0) x 4 bytes under MAX_TENSOR_BYTES; one range = no split."""
    per = max(per_image_elems) * 4
    if B * per < MAX_TENSOR_BYTES or B == 1:
        return [(0, B)]
    step = max(1, int((MAX_TENSOR_BYTES - 1) // per))
    return [(b, min(B, b + step)) for b in range(0, B, step)]


def conv2d_fwd(x, w, bias, stride, pad, drop_p=0.0, drop_key=0, dil=1, w_amax=None):
    B, H, W, Cin = x.shape
    Cout, Cin2, R, S = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    Ho, Wo = _out_hw(H, W, R, S, stride, pad, dil)
    ranges = _batch_ranges(B, H * W * Cin, Ho * Wo * Cout) if drop_p == 0.0 else [(0, B)]
    if len(ranges) > 1:
        return torch.cat([conv2d_fwd(x[b0:b1], w, bias, stride, pad, 0.0, 0, dil, w_amax) for b0, b1 in ranges])
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    kind = _conv_kind(Cin, Cout, R, S, stride, pad, dil)
    ws, ws_bytes, packed = _pack_for(w, x.device, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0)
    _launch_conv(packed, kind, 2.0 * B * Ho * Wo * Cout * R * S * Cin, "fs_conv2d_fwd", hip.ptr(x), hip.ptr(rsck(w)), hip.ptr(bias),
            hip.ptr(y), B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, float(drop_p), int(drop_key), hip.ptr(ws), ws_bytes,
            hip.ptr(w_amax))
    return y


# ---- weight packs that outlive the call ------------------------------------------------------------------------------------------
# Every split-precision conv kernel starts with a small launch that writes the layer's weights, split into their 16-bit terms in
# consumption order, into its scratch (438 + 186 such launches per HRNet step).  The library can keep a pack across calls
# (fs_conv2d_pack / fs_conv2d_ws_mode): a parameter whose owner says when it changes -- an optimiser arena (train.FlatParams bumps
# its epoch word at FlatAdam.step, a broadcast, load_state_dict) or static_weight_packs(module) for a serving process -- then keeps one
# scratch per (problem, direction, precision mode); with an arena, `train_step` refills the scratches used in the last step on a side
# stream right after the optimiser step (repack_weights) and the conv entry points are told the scratch is already packed.
# Measured on the headline step (profiles/r04/pack_persist_ab.txt): +-0 (374.7 / 372.7 / 372.9 -> 374.6 / 373.5 / 372.7 img/s); with NO
# pack launch at all (stale packs, timing only) 172.0 -> 170.7 ms -- the 4.5 ms the packs take when the kernels are serialised is
# hidden in the real step, and packs on a side stream still need the CUs the persistent conv kernels fill.  configs[4] +1-2 %,
# configs[3] -0.7 % (19 ms more host time per step).  So: OFF for arena parameters unless FS_PACK_PERSIST=1 (read once); always on
# for parameters marked static.  Parameters with neither (bare tensors in tests, derived weights built per call) pack per call.
PACK_PERSIST = os.environ.get("FS_PACK_PERSIST", "0") == "1"
PACK_GROUP = 32           # packs per hand-over event of the prefetch
_PACK_ORDER = []          # weak references to every live pack, in first-use order (= the order the next step needs them)
_PACK_SIDE = {}           # device index -> torch Stream of the prefetch


class _PackGroup:
    __slots__ = ("event", "waited")

    def __init__(self):
        self.event, self.waited = None, set()


class _Pack:
    __slots__ = ("w", "args", "transposed", "ws", "n", "mode", "cell", "epoch", "version", "used", "group", "stream", "ready", "__weakref__")


def _pack_for(w, device, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed):
    """(scratch | None, bytes, already packed?) for a conv entry point on weight tensor `w`."""
    n = hip.conv_workspace_bytes(H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed)
    if n == 0:
        return None, 0, False
    cell = getattr(w, "_fs_epoch", None) if (PACK_PERSIST or getattr(w, "_fs_static", False)) else None
    if cell is None or not hip.query("fs_conv2d_pack_persistent", B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed, n):
        return torch.empty(n, device=device, dtype=torch.uint8), n, False
    packs = w.__dict__.get("_fs_packs")
    if packs is None:
        packs = w.__dict__["_fs_packs"] = {}
    mode = hip.get_conv_precision()
    key = (transposed, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, mode)
    e = packs.get(key)
    if e is None:
        e = packs[key] = _Pack()
        e.w, e.args, e.transposed, e.n, e.mode = weakref.ref(w), key[1:13], transposed, n, mode
        e.ws = torch.empty(n, device=device, dtype=torch.uint8)
        e.cell, e.epoch, e.version, e.group, e.stream, e.ready = cell, -1, -1, None, None, set()
        _PACK_ORDER.append(weakref.ref(e))
    e.used = True
    so = hip.stream_override()
    cur = so if so is not None else hip._stream()
    if (e.cell is cell and e.epoch == cell[0] and e.version == w._version):
        if cur not in e.ready:          # first use on this stream since the pack was enqueued: order the stream behind it
            g = e.group
            if g is not None:
                if cur not in g.waited:
                    # the wait goes on `cur`, the stream the launch uses (which is the thread's stream override when that is set), not on torch's
                    # current stream
                    (torch.cuda.current_stream() if cur == hip._stream() else torch.cuda.ExternalStream(cur)).wait_event(g.event)
                    g.waited.add(cur)
            elif e.stream is not None and e.stream != cur:
                hip.stream_wait(cur, e.stream)
            e.ready.add(cur)
        return e.ws, n, True
    # stale (or new): this call packs on its own stream, later calls of the same epoch run on the result.  The scratch is rewritten IN
    # PLACE: consumers of the previous pack on other streams (HRNet branch streams, the autograd streams of the bwd-data packs) must
    # have finished reading it first
    for s_prev in e.ready:
        if s_prev != cur:
            hip.stream_wait(cur, s_prev)
    if e.stream is not None and e.stream != cur and e.stream not in e.ready:
        hip.stream_wait(cur, e.stream)
    e.cell, e.epoch, e.version, e.group, e.stream, e.ready = cell, cell[0], w._version, None, cur, {cur}
    return e.ws, n, False


def static_weight_packs(module, on=True):
    """Serving: the parameters of `module` do not change between forwards (no optimiser): keep every conv layer's weight pack after its
    first forward -- no pack launch from the second forward on.  Call again after rewriting the weights through anything that does not
    move torch's version counters; load_state_dict and in-place torch ops are seen without it."""
    for p in module.parameters():
        if on:
            cell = p.__dict__.get("_fs_epoch")
            if cell is None:
                p._fs_epoch = [0]
            else:
                cell[0] += 1
            p._fs_static = True
        else:
            p.__dict__.pop("_fs_static", None)
            p.__dict__.pop("_fs_packs", None)


def repack_weights():
    """Refill, on a side stream, every weight pack that was used since the last call and whose arena has been rewritten since it was
    packed (train.train_step calls this right after the optimiser steps; a frozen optimiser's packs stay valid and are skipped).  The
    consuming streams wait for the event of the pack's group at first use (_pack_for)."""
    if not _PACK_ORDER or TIMER is not None:
        return
    mode = hip.get_conv_precision()
    todo, live = [], []
    for r in _PACK_ORDER:
        e = r()
        if e is None:
            continue
        live.append(r)
        w = e.w()
        if w is None or not e.used or e.mode != mode:
            continue
        e.used = False
        cell = getattr(w, "_fs_epoch", None)
        if cell is None or (e.cell is cell and e.epoch == cell[0] and e.version == w._version):
            continue
        todo.append((e, w, cell))
    _PACK_ORDER[:] = live
    if not todo:
        return
    dev = todo[0][1].device
    side = _PACK_SIDE.get(dev.index)
    if side is None:
        side = _PACK_SIDE[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())          # the optimiser step (and the arena's max|w| refresh) come first
    raw = side.cuda_stream
    hip.set_stream_override(raw)
    try:
        group = _PackGroup()
        for i, (e, w, cell) in enumerate(todo):
            hip.call("fs_conv2d_pack", hip.ptr(rsck(w)), *e.args, e.transposed, hip.ptr(e.ws), e.n, hip.ptr(weight_amax(w)))
            e.cell, e.epoch, e.version, e.group, e.stream, e.ready = cell, cell[0], w._version, group, raw, set()
            if (i + 1) % PACK_GROUP == 0 or i + 1 == len(todo):
                group.event = torch.cuda.Event()
                group.event.record(side)
                group = _PackGroup()
    finally:
        hip.set_stream_override(None)


def _launch_conv(packed, kind, flops, name, *args):
    """_launch for the conv entry points; packed = their scratch already holds the weight pack (fs_conv2d_ws_mode)."""
    if not packed:
        return _launch(kind, flops, name, *args)
    if TIMER is None:
        hip.call_packed(name, *args)
    else:
        TIMER.launch(kind, flops, lambda: hip.call_packed(name, *args), tag=(name,) + tuple(a for a in args if type(a) is int and a < (1 << 20)))


FUSE_BN_STATS = True     # BatchNorm batch statistics come out of the conv epilogue (fs_conv2d_fwd_stats)
FUSE_EVAL_BN = True      # inference: eval-mode BatchNorm + residual + activation in the conv epilogue (fs_conv2d_fwd_affine_act)
# BatchNorm-backward column sums come out of the kernel that produces the gradient, where one does (FanOut's add, the F(2,3) bwd-data
# epilogue); FS_FUSE_BN_BWD=0 (read once, A/B runs) keeps every layer on its own reduction pass
FUSE_BN_BWD_SUMS = os.environ.get("FS_FUSE_BN_BWD", "1") != "0"


def conv2d_fwd_stats(x, w, bias, stride, pad, drop_p=0.0, drop_key=0, dil=1, w_amax=None):
    """Forward conv + per-workgroup BatchNorm partial sums (slab [nwg][Cout][2])."""
    B, H, W, Cin = x.shape
    Cout, Cin2, R, S = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    Ho, Wo = _out_hw(H, W, R, S, stride, pad, dil)
    ranges = _batch_ranges(B, H * W * Cin, Ho * Wo * Cout) if drop_p == 0.0 else [(0, B)]
    if len(ranges) > 1:          # slab rows of the ranges concatenate: the finalize pass sums over all of them
        parts = [conv2d_fwd_stats(x[b0:b1], w, bias, stride, pad, 0.0, 0, dil, w_amax) for b0, b1 in ranges]
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), sum(p[2] for p in parts)
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    ws, ws_bytes, packed = _pack_for(w, x.device, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0)
    nwg = hip.conv_stats_slabs(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes)
    slab = torch.empty(nwg * Cout * 2, device=x.device, dtype=torch.float32)
    _launch_conv(packed, _conv_kind(Cin, Cout, R, S, stride, pad, dil), 2.0 * B * Ho * Wo * Cout * R * S * Cin, "fs_conv2d_fwd_stats", hip.ptr(x), hip.ptr(rsck(w)),
            hip.ptr(bias), hip.ptr(y), hip.ptr(slab), B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, float(drop_p), int(drop_key),
            hip.ptr(ws), ws_bytes, hip.ptr(w_amax))
    return y, slab, nwg


def _unmask_bits(amask, like):
    """1 byte per 4 channels -> float {0,1} tensor shaped like `like` (fallback path only)."""
    bits = (amask.view(-1, 1) >> torch.arange(4, device=amask.device, dtype=torch.uint8)) & 1
    return bits.reshape(like.shape).to(like.dtype)


def conv2d_bwd_data(dy, w, x_shape, stride, pad, dil=1, w_amax=None, src_bn=None, addend=None):
    """dX of conv2d.  src_bn = the `_fs_bn` record of the layer that PRODUCED x (x = its output z, this conv its consumer): where the
    kernel this shape runs on can, its epilogue also forms that layer's BatchNorm-backward column sums and the slab is left in BN_SLABS
    under dx's address for ConvBnAct.backward to pick up (if autograd adds another consumer's gradient to dx the sum is a new tensor,
    the lookup misses and the layer runs its own reduction pass).  addend = (dz, mask bytes | None): a second gradient of x -- the
    residual branch's -- added in the same epilogue (ConvBnAct.backward stashes it in PENDING_RES only after asking the library that
    this shape's kernel takes it)."""
    B, H, W, Cin = x_shape
    Cout, _, R, S = w.shape
    _, Ho, Wo, _ = dy.shape
    ranges = _batch_ranges(B, H * W * Cin, Ho * Wo * Cout)
    if len(ranges) > 1:          # without the fused extras; the addend joins afterwards
        dx = torch.cat([conv2d_bwd_data(dy[b0:b1], w, (b1 - b0, H, W, Cin), stride, pad, dil, w_amax) for b0, b1 in ranges])
        if addend is not None:
            a_src, a_mask = addend
            dx += a_src if a_mask is None else a_src * _unmask_bits(a_mask, a_src)
        return dx
    dx = torch.empty(B, H, W, Cin, device=dy.device, dtype=torch.float32)
    kind = _conv_kind(Cout, Cin, R, S, stride, pad, dil)
    ws, ws_bytes, packed = _pack_for(w, dy.device, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 1)
    flops = 2.0 * B * Ho * Wo * Cout * R * S * Cin
    if src_bn is not None and not (FUSE_BN_BWD_SUMS and FANOUT and tuple(src_bn[0].shape) == (B, H, W, Cin)):
        src_bn = None
    if src_bn is not None or addend is not None:
        rows = hip.bwd_data_bnsum_slabs(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes)
        if rows > 0:
            y, amask, mean, invstd = src_bn[:4] if src_bn is not None else (None, None, None, None)
            slab = torch.empty(rows * Cin * 2, device=dy.device, dtype=torch.float32) if src_bn is not None else None
            a_src, a_mask = addend if addend is not None else (None, None)
            _launch_conv(packed, kind, flops, "fs_conv2d_bwd_data_bnsum", hip.ptr(dy), hip.ptr(rsck(w)), hip.ptr(dx), B, H, W, Cin, Ho, Wo, Cout, R, S,
                    stride, pad, dil, hip.ptr(ws), ws_bytes, hip.ptr(w_amax), hip.ptr(y), hip.ptr(amask), hip.ptr(mean), hip.ptr(invstd),
                    hip.ptr(slab), hip.ptr(a_src), hip.ptr(a_mask))
            if slab is not None:
                BN_SLABS[(dx.data_ptr(), y.data_ptr())] = (slab, rows, dx)
            return dx
    _launch_conv(packed, kind, flops, "fs_conv2d_bwd_data", hip.ptr(dy), hip.ptr(rsck(w)), hip.ptr(dx),
            B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, hip.ptr(ws), ws_bytes, hip.ptr(w_amax))
    if addend is not None:          # the library no longer takes the addend for this shape (precision mode changed since the stash): add here
        a_src, a_mask = addend
        dx = dx + (a_src if a_mask is None else a_src * _unmask_bits(a_mask, a_src))
    return dx


# Weight / affine gradients of a parameter whose `.grad` IS its slice of a flat gradient arena (train.FlatParams registers the
# slice as `p._fs_grad_home`) are ADDED straight into that slice by the kernels (conv weights and BatchNorm affines alike) and
# the autograd function returns None for them: no temporary, no AccumulateGrad add.  The decision is per parameter -- there is
# no process-wide mode: a parameter outside any arena, or whose `.grad` was re-pointed (module.zero_grad() -> None -> a fresh
# tensor), takes the ordinary autograd path.  The arena is zeroed once per step (optimizer.zero_grad), so repeated backwards
# accumulate exactly like `.grad`.  DIRECT_GRAD = False (or FS_DIRECT_GRAD=0) switches the direct path off altogether.
#
# torch's DistributedDataParallel (the reference's call site, train_deform_semantic.py:395) learns that a gradient is ready from a
# hook on the parameter's AccumulateGrad node, which a kernel that adds straight into the arena never fires.  A forward that runs
# inside a DDP wrapper therefore suspends the direct path for itself and its backward (models.DeformSegmentationModule.forward
# -> under_torch_ddp): gradients are returned to autograd, AccumulateGrad adds them into the same arena views in place, the
# reducer sees every one of them.
DIRECT_GRAD = os.environ.get("FS_DIRECT_GRAD", "1") != "0"
DDP_ACTIVE = False       # set per forward by models.DeformSegmentationModule.forward


def under_torch_ddp(module) -> bool:
    """True while `module.forward` is being run by a torch DistributedDataParallel wrapper around `module`."""
    ddp_cls = torch.nn.parallel.DistributedDataParallel
    active = getattr(ddp_cls, "_active_ddp_module", None)
    return active is not None and getattr(active, "module", None) is module


PARAM_VIEW_DIRECT = os.environ.get("FS_PARAM_VIEW_DIRECT", "1") != "0"


def param_view(leaf, fn):
    """fn(leaf): a VIEW of a parameter over the same storage (nn.Linear's (out, in) weight as the (out, in, 1, 1) filter of a 1x1 conv, a
    k x k stride-k filter as the (k*k*Cin, Cout) matrix of its patch rows) that remembers how it was made, so that a kernel writing the
    view's gradient can write it through the same view of the parameter's arena slice (_direct_grad_target) -- without that the gradient
    is a temporary that travels back through ViewBackward to an AccumulateGrad add: 370 small launches per configs[3] step."""
    v = fn(leaf)
    if PARAM_VIEW_DIRECT:
        v._fs_grad_via = (leaf, fn)
    return v


def _direct_grad_target(p):
    via = getattr(p, "_fs_grad_via", None)
    if via is not None:
        g = _direct_grad_target(via[0])
        if g is None:
            return None
        t = via[1](g)
        return t if (t.data_ptr() == g.data_ptr() and t.shape == p.shape and t.stride() == p.stride()) else None
    home = getattr(p, "_fs_grad_home", None)
    if not DIRECT_GRAD or DDP_ACTIVE or home is None or not p.is_leaf:
        return None
    g = p.grad
    if g is not None and g.is_cuda and g.data_ptr() == home[0].data_ptr() + 4 * home[1] and g.shape == p.shape and g.stride() == p.stride():
        return g
    return None


def conv2d_bwd_weight(x, dy, w_shape, stride, pad, out=None, dil=1, accumulate=False, keep=None):
    """dW of conv2d.  out = a gradient buffer in the weight's layout; accumulate=True adds into it (the DIRECT_GRAD arena
    is zeroed once per step by zero_grad, so the per-layer memset is skipped and repeated backwards accumulate like .grad)."""
    B, H, W, Cin = x.shape
    Cout, _, R, S = w_shape
    _, Ho, Wo, _ = dy.shape
    ranges = _batch_ranges(B, H * W * Cin, Ho * Wo * Cout)
    if len(ranges) > 1:          # the ranges' gradients accumulate in one buffer
        buf = out if out is not None else new_rsck_weight(Cout, Cin, R, S, device=x.device)
        for i, (b0, b1) in enumerate(ranges):
            conv2d_bwd_weight(x[b0:b1], dy[b0:b1], w_shape, stride, pad, out=buf, dil=dil, accumulate=accumulate or i > 0)
        return buf
    dw = rsck(out) if out is not None else torch.empty(R, S, Cin, Cout, device=x.device, dtype=torch.float32)
    k3 = R == 3 and S == 3 and stride == 1 and pad == 1 and dil == 1 and Cin % 4 == 0 and Cout % 4 == 0 and Cin >= 16 and Cout >= 16
    dws, dws_bytes = hip.wgrad_workspace(x.device, Cin, Cout, R, S, stride, pad, dil)      # deterministic mode; strided 3x3 layers
    if keep is not None and dws is not None:
        keep.append(dws)          # the launch runs on a side stream: the scratch must outlive this call (join_wgrad_streams)
    _launch("wgrad3x3" if (k3 and hip.get_conv_precision() != "f32") else "conv_wgrad", 2.0 * B * Ho * Wo * Cout * R * S * Cin,
            "fs_conv2d_bwd_weight", hip.ptr(x), hip.ptr(dy), hip.ptr(dw),
            B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 1 if accumulate else 0, hip.ptr(dws), dws_bytes)
    return out if out is not None else dw.permute(3, 2, 0, 1)


def colsum(x2d_rows, C, into=None):
    """Column sums (bias gradients).  into = a gradient-arena target: the sums are ADDED to it and None is returned."""
    M = x2d_rows.numel() // C
    scratch = torch.empty(hip.query("fs_colsum_scratch_floats", M, C), device=x2d_rows.device, dtype=torch.float32)
    if into is not None:
        hip.call("fs_colsum", hip.ptr(x2d_rows), M, C, hip.ptr(into), 1, hip.ptr(scratch))
        return None
    out = torch.empty(C, device=x2d_rows.device, dtype=torch.float32)
    hip.call("fs_colsum", hip.ptr(x2d_rows), M, C, hip.ptr(out), 0, hip.ptr(scratch))
    return out


# ----------------------------------------------------------------------------------------------
# Layers fed by 3 or 5 input channels (HRNet stem, saliency net) as aligned problems: the generic scalar-channel kernels spend
# 0.4-1.3 ms per launch on them where the aligned ones are bound by the 105-315 MB of activations they move.
# ----------------------------------------------------------------------------------------------
PAD_ODD_CHANNELS = os.environ.get("FS_PAD_ODD_CHANNELS", "1") != "0"


def padded_in_channels(cin):
    """Input-channel count the aligned kernels accept for a layer with `cin` channels (bwd-weight wants >= 16, all want x4)."""
    return 16 if cin < 16 else (cin + 3) // 4 * 4


class PadWeightChannels(Function):
    """(Cout,Cin,R,S) RSCK weight -> (Cout,Cp,R,S) RSCK weight with zero taps for the padding channels; backward slices."""

    @staticmethod
    def forward(ctx, w, cp):
        cout, cin, r, s = w.shape
        wp = new_rsck_weight(cout, cp, r, s, device=w.device)
        v = rsck(wp)
        v.zero_()
        v[:, :, :cin, :].copy_(rsck(w))
        ctx.cin = cin
        return wp

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.cin], None


class CenterTap(Function):
    """(Cout,Cin,3,3) RSCK weight -> its centre tap as a (Cout,Cin,1,1) RSCK weight; backward = zeros with the centre filled in."""

    @staticmethod
    def forward(ctx, w):
        cout, cin, r, s = w.shape
        wc = new_rsck_weight(cout, cin, 1, 1, device=w.device)
        rsck(wc)[0, 0].copy_(rsck(w)[r // 2, s // 2])
        ctx.rs = (r, s)
        return wc

    @staticmethod
    def backward(ctx, g):
        r, s = ctx.rs
        cout, cin = g.shape[:2]
        gw = new_rsck_weight(cout, cin, r, s, device=g.device)
        v = rsck(gw)
        v.zero_()
        v[r // 2, s // 2].copy_(rsck(g)[0, 0])
        return gw


def pad_in_channels(x, w):
    """(x (B,H,W,Cin), w) -> (x padded with zero channels, w padded alike) when Cin is not a multiple of 4 and the split-precision
    kernels are on; exact: the extra products are 0 * 0."""
    cin = w.shape[1]
    if not PAD_ODD_CHANNELS or cin % 4 == 0 or w.shape[2] * w.shape[3] > 32 or hip.get_conv_precision() == "f32":
        return x, w          # aligned already, or a filter beyond the aligned kernels' 32 taps (7x7 ResNet stem): generic kernels
    cp = padded_in_channels(cin)
    return torch.nn.functional.pad(x, (0, cp - cin)), PadWeightChannels.apply(w, cp)


FANOUT = os.environ.get("FS_FANOUT", "1") != "0"

# Launch-latency-bound layers (DeepLab's 10x10 / 20x20 stages at 16 images per GPU: ~1 700 launches of 5-15 us per 30 ms step, the GPU
# mostly waiting on the dependency chain): a weight gradient has no consumer before the optimiser, so below WGRAD_SIDE_FLOPS it is
# launched on a side stream beside the bwd-data chain.  Only where the kernel adds straight into the gradient arena (no autograd consumer);
# FlatAdam.step / zero_grad and train.allreduce_gradients join the side stream first (join_wgrad_streams).  The headline layers are 30 GFLOP
# each and keep the one-stream order: there the GPU is saturated and a side stream was measured at +-0.
WGRAD_SIDE_FLOPS = float(os.environ.get("FS_WGRAD_SIDE_GFLOP", "4")) * 1e9
_WGRAD_SIDE = {}            # device index -> (torch Stream, raw handle)
_WGRAD_SIDE_BUSY = {}       # raw handle of a side stream with launches since the last join -> tensors those launches read or write
                            # (kept alive while the side stream may still use them: they were allocated on the launching stream's pool,
                            # so freeing them earlier would let the allocator hand them out again; cheaper than three record_stream calls)
_WGRAD_RETIRED = []         # (event recorded on the side stream, tensors): kept lists that were joined or rotated out; dropped once the
                            # event has COMPLETED -- a join only enqueues a wait, the side stream may still be reading the tensors
_WGRAD_ROTATE = 64          # side launches per kept list before it is rotated out (bounds what a long backward holds on to)
_WGRAD_JOIN_QUEUED = [False]


def _wgrad_side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _WGRAD_SIDE.get(key)
    if s is None:
        st = torch.cuda.Stream(device=device)
        s = _WGRAD_SIDE[key] = (st, st.cuda_stream)
    return s[1]


def _wgrad_stream_obj(raw):
    for st, h in _WGRAD_SIDE.values():
        if h == raw:
            return st
    return None


def _wgrad_retire(raw, keep):
    """The kept tensors of side stream `raw` may be dropped once everything enqueued there so far has run: remember them under an event.
    (Under stream capture nothing runs and an event cannot be queried; the caller drops the list at the join instead -- every node
    captured behind the join is ordered behind the side launches, and the graph's private pool keeps the addresses for its replays.)"""
    st = _wgrad_stream_obj(raw)
    if st is None or not keep or torch.cuda.is_current_stream_capturing():
        return
    ev = torch.cuda.Event()
    ev.record(st)
    _WGRAD_RETIRED.append((ev, keep))


def _wgrad_reap():
    if _WGRAD_RETIRED and not torch.cuda.is_current_stream_capturing():
        _WGRAD_RETIRED[:] = [(ev, keep) for ev, keep in _WGRAD_RETIRED if not ev.query()]


def join_wgrad_streams():
    """The current stream waits for every weight-gradient launch that went to a side stream since the last join.  Since round 5 this runs
    as a final callback of the autograd engine at the end of every backward that used the side stream (_queue_wgrad_join), i.e. with the
    CALLER's current stream -- the stream backward() returns on -- so any reader of .grad / the gradient arena after backward() is ordered
    behind the side stream like behind every other gradient.  FlatAdam.step / zero_grad and train.allreduce_gradients still call it as a
    backstop (no-ops then)."""
    if _WGRAD_SIDE_BUSY:
        cur = hip._stream()
        for h, keep in _WGRAD_SIDE_BUSY.items():
            hip.stream_wait(cur, h)
            _wgrad_retire(h, keep)
        _WGRAD_SIDE_BUSY.clear()
    _wgrad_reap()


def _wgrad_join_callback():
    _WGRAD_JOIN_QUEUED[0] = False
    join_wgrad_streams()


def _queue_wgrad_join():
    """First side launch of a backward pass: have the engine run the join when the pass ends (torch runs final callbacks after it has
    ordered the caller's stream behind the leaf streams, with the caller's current streams set)."""
    if _WGRAD_JOIN_QUEUED[0]:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join_callback)
        _WGRAD_JOIN_QUEUED[0] = True
    except RuntimeError:          # not inside an engine run (a Function.backward called by hand): the explicit joins remain
        pass


# BatchNorm-backward column sums that the PRODUCER of a gradient tensor already formed (FanOut.backward below, the bwd-data epilogue of the
# consumer conv, HrFuse.backward): (dz.data_ptr(), y.data_ptr()) -> (slab, nslab, dz), y = the conv output of the layer the sums are for
# (one fuse gradient is the output gradient of up to three layers).  The entry keeps dz alive, so its address cannot be reused while the
# entry exists; ConvBnAct.backward pops it.  Cleared at the start of every module forward.
BN_SLABS = {}


class FanOut(Function):
    """x -> n aliases of x, one per consumer.  Backward: ONE n-ary HIP add of the consumers' gradients (fs_add_n, up to four at a
    time) instead of the n - 1 binary ATen adds the autograd engine would issue -- the last ATen compute kernel inside the step.
    When x is the output of a conv + BatchNorm + activation layer (ConvBnAct tags it with `_fs_bn`), the last add also forms that
    layer's BatchNorm-backward column sums (fs_add_n_bnsum): the layer's own reduction pass over (dz, y) is then skipped."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.bn = getattr(x, "_fs_bn", None)
        ctx.set_materialize_grads(False)      # a consumer whose gradient was absorbed elsewhere returns None: keep it None, not a zero tensor
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g if g.is_contiguous() else g.contiguous() for g in grads if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        rest = gs[1:]
        if rest and (acc.numel() % 4 or not acc.is_cuda):
            for g in rest:
                acc = acc + g
            return acc, None
        bn = ctx.bn if FUSE_BN_BWD_SUMS else None
        while rest:
            take, rest = rest[:3], rest[3:]
            out = torch.empty_like(acc)
            if bn is not None and not rest and acc.dim() == 4 and acc.shape[-1] % 4 == 0 and acc.shape == bn[0].shape:
                y, amask, mean, invstd, act = bn
                C = acc.shape[-1]
                M = acc.numel() // C
                nslab = hip.bn_bwd_slabs(M, C)
                slab = torch.empty(nslab * C * 2, device=acc.device, dtype=torch.float32)
                hip.call("fs_add_n_bnsum", hip.ptr(acc), hip.ptr(take[0]), hip.ptr(take[1]) if len(take) > 1 else None,
                         hip.ptr(take[2]) if len(take) > 2 else None, hip.ptr(out), hip.ptr(amask), hip.ptr(y), hip.ptr(mean),
                         hip.ptr(invstd), M, C, act if amask is not None else ACT_NONE, hip.ptr(slab))
                BN_SLABS[(out.data_ptr(), y.data_ptr())] = (slab, nslab, out)
            else:
                hip.call("fs_add_n", hip.ptr(acc), hip.ptr(take[0]), hip.ptr(take[1]) if len(take) > 1 else None,
                         hip.ptr(take[2]) if len(take) > 2 else None, hip.ptr(out), acc.numel())
            acc = out
        return acc, None


_FAN_IDS = [0]
# A two-way fan-out whose consumers are a convolution and a residual input (BasicBlock, models/hrnetv2_nodownsp.py:46-62): the conv
# consumer's forward leaves its geometry in FAN_GEOM under the fan id, the residual consumer's backward (it always runs first) asks the
# library whether that convolution's bwd-data epilogue takes an addend and, if so, leaves (dz, mask bytes) in PENDING_RES instead of
# materialising the residual gradient; the conv consumer's backward passes it on (conv2d_bwd_data) -- no n-ary add, no dres tensor.
FAN_GEOM = {}
PENDING_RES = {}
FAN_DONE = set()        # fan ids whose conv consumer has run its backward


def reset_step_state():
    """Start of a module forward: drop the per-step records.  A residual gradient still waiting in PENDING_RES was never added to anything
    -- a silently wrong gradient -- so that is an error, not something to clear."""
    if PENDING_RES:
        left = sorted(PENDING_RES)
        PENDING_RES.clear()
        raise hip.HipLibraryError(f"residual gradients of fan-outs {left} were stashed for a convolution's bwd-data epilogue that never ran")
    BN_SLABS.clear()
    FAN_GEOM.clear()
    FAN_DONE.clear()
    _WGRAD_JOIN_QUEUED[0] = False      # (a backward that raised never ran its final callbacks)


def fan_out(x, n):
    """n references to x for n consumers (n >= 2 and FS_FANOUT on: through FanOut, else x itself n times)."""
    if n < 2 or not FANOUT or not x.requires_grad:
        return (x,) * n
    outs = FanOut.apply(x, n)
    if n == 2:
        _FAN_IDS[0] += 1
        tag = (_FAN_IDS[0], getattr(x, "_fs_bn", None))
        for o in outs:
            o._fs_fan = tag
    return outs


class StashGrad(Function):
    """Identity on one alias of a two-way fan-out whose OTHER alias feeds a convolution (C1: `feat` -> the 3x3 conv of the mask branch and ->
    the classification branch).  Backward: the gradient that arrives here is left in PENDING_RES under the fan id and None is returned, so the
    convolution's bwd-data adds it in its epilogue (one read) instead of FanOut.backward adding two 1.57 GB tensors in a pass of its own
    (806 us per configs[1] step).  The branch behind this node must run its backward BEFORE the convolution's: put the node in the forward
    AFTER the convolution branch (the engine runs ready nodes in reverse order of creation); if the convolution's backward came first anyway
    (FAN_DONE), the gradient is returned as usual and the fan-out adds it."""

    @staticmethod
    def forward(ctx, x, fan_id):
        ctx.fan_id = fan_id
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if g is None or ctx.fan_id in FAN_DONE or ctx.fan_id in PENDING_RES or not (FUSE_BN_BWD_SUMS and FANOUT):
            return g, None
        PENDING_RES[ctx.fan_id] = (g if g.is_contiguous() else g.contiguous(), None)
        return None, None


FANOUT_SUBSAMPLE = os.environ.get("FS_FANOUT_SUBSAMPLE", "1") != "0"


class FanOutSubsample(Function):
    """x (B,H,W,C) -> (x, x[:, ::s, ::s, :] contiguous).  Backward: the subsampled branch's gradient is added in place into the
    full-resolution branch's gradient at the sampled pixels (that tensor is the fresh dX of the branch's first consumer)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        ctx.shape = x.shape
        return x.view_as(x), x[:, ::s, ::s, :].contiguous()

    @staticmethod
    def backward(ctx, g_full, g_sub):
        s = ctx.s
        if g_full is None:
            g_full = torch.zeros(ctx.shape, device=g_sub.device, dtype=g_sub.dtype)
        elif not g_full.is_contiguous():
            g_full = g_full.contiguous()
        if g_sub is not None:
            g_full[:, ::s, ::s, :].add_(g_sub)
        return g_full, None


def act_has_bwd(act, amask):
    """The activation derivative is available to another kernel: no activation, or the 1-byte masks were written."""
    return act == ACT_NONE or amask is not None


# ----------------------------------------------------------------------------------------------
# conv (+bias) (+dropout) + BatchNorm + residual + activation
# ----------------------------------------------------------------------------------------------
class ConvBnAct(Function):
    """z = act(BN(dropout(conv(x, w) + bias)) + res).
    meta: dict(stride, pad, act, training, momentum, drop_p, drop_key, running_mean, running_var,
    num_batches_tracked)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, res, meta):
        training = meta["training"]
        drop_p = meta["drop_p"] if training else 0.0
        Cout, Cin = w.shape[0], w.shape[1]
        fused_stats = training and FUSE_BN_STATS and Cin % 4 == 0 and Cout % 4 == 0
        dil = meta.get("dil", 1)
        wa = weight_amax(w)
        # (ctx.needs_input_grad mirrors requires_grad of the inputs even under no_grad: "no backward will follow" = grad mode was off at
        #  the call site, which modules.conv_bn_act records in meta before Function.apply switches it off)
        if not training and FUSE_EVAL_BN and not meta.get("grad_enabled", True) and Cin % 4 == 0 and Cout % 4 == 0:
            # inference: eval-mode BatchNorm, residual and activation in the conv epilogue where this shape's kernel has one
            B_, H_, W_, _ = x.shape
            R_, S_ = w.shape[2], w.shape[3]
            Ho_, Wo_ = _out_hw(H_, W_, R_, S_, meta["stride"], meta["pad"], dil)
            ws_bytes = hip.conv_workspace_bytes(H_, W_, Cin, Ho_, Wo_, Cout, R_, S_, meta["stride"], meta["pad"], dil, 0)
            if hip.fwd_affine_act_ok(B_, H_, W_, Cin, Ho_, Wo_, Cout, R_, S_, meta["stride"], meta["pad"], dil, ws_bytes):
                coef = torch.empty(2, Cout, device=x.device, dtype=torch.float32)
                hip.call("fs_bn_eval_affine", hip.ptr(meta["running_mean"]), hip.ptr(meta["running_var"]), hip.ptr(gamma), hip.ptr(beta), Cout,
                         BN_EPS, hip.ptr(coef[0]), hip.ptr(coef[1]))
                z = torch.empty(B_, Ho_, Wo_, Cout, device=x.device, dtype=torch.float32)
                ws, ws_bytes, packed = _pack_for(w, x.device, B_, H_, W_, Cin, Ho_, Wo_, Cout, R_, S_, meta["stride"], meta["pad"], dil, 0)
                _launch_conv(packed, _conv_kind(Cin, Cout, R_, S_, meta["stride"], meta["pad"], dil), 2.0 * B_ * Ho_ * Wo_ * Cout * R_ * S_ * Cin,
                        "fs_conv2d_fwd_affine_act", hip.ptr(x), hip.ptr(rsck(w)), hip.ptr(bias), hip.ptr(coef[0]), hip.ptr(coef[1]), hip.ptr(res),
                        hip.ptr(z), B_, H_, W_, Cin, Ho_, Wo_, Cout, R_, S_, meta["stride"], meta["pad"], dil, meta["act"], hip.ptr(ws), ws_bytes,
                        hip.ptr(wa))
                return z
        if fused_stats:
            y, slab, nwg = conv2d_fwd_stats(x, w, bias, meta["stride"], meta["pad"], drop_p, meta["drop_key"], dil, w_amax=wa)
        else:
            y = conv2d_fwd(x, w, bias, meta["stride"], meta["pad"], drop_p, meta["drop_key"], dil, w_amax=wa)
        B, Ho, Wo, C = y.shape
        M = B * Ho * Wo
        mean = torch.empty(C, device=y.device, dtype=torch.float32)
        invstd = torch.empty(C, device=y.device, dtype=torch.float32)
        if fused_stats:
            hip.call("fs_bn_finalize_slab", hip.ptr(slab), nwg, M, C, float(meta["momentum"]), BN_EPS,
                     hip.ptr(meta["running_mean"]), hip.ptr(meta["running_var"]), hip.ptr(mean), hip.ptr(invstd))
            meta["num_batches_tracked"].add_(1)
        elif training:
            sums = torch.empty(hip.query("fs_bn_stats_scratch_doubles", M, C), device=y.device, dtype=torch.float64)
            hip.call("fs_bn_stats", hip.ptr(y), M, C, float(meta["momentum"]), BN_EPS, hip.ptr(meta["running_mean"]),
                     hip.ptr(meta["running_var"]), hip.ptr(mean), hip.ptr(invstd), hip.ptr(sums))
            meta["num_batches_tracked"].add_(1)
        else:
            hip.call("fs_bn_eval_prepare", hip.ptr(meta["running_mean"]), hip.ptr(meta["running_var"]), C, BN_EPS,
                     hip.ptr(mean), hip.ptr(invstd))
        z = torch.empty_like(y)
        # activation-derivative bits for the backward passes (1 byte per 4 channels instead of re-reading z)
        amask = torch.empty(M * C // 4, device=y.device, dtype=torch.uint8) if (meta["act"] != 0 and any(ctx.needs_input_grad)) else None
        # timer "work" of the HBM-bound kinds = algorithmic bytes: every operand read once, every result written once
        _launch("bn_fwd", 4.0 * M * C * (2 + (res is not None)) + (M * C // 4 if amask is not None else 0),
                "fs_bn_act_fwd", hip.ptr(y), hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(beta), hip.ptr(res),
                hip.ptr(z), hip.ptr(amask), M, C, meta["act"])
        ctx.meta = dict(stride=meta["stride"], pad=meta["pad"], dil=dil, act=meta["act"], training=training, drop_p=drop_p,
                        drop_key=meta["drop_key"], has_bias=bias is not None, has_res=res is not None)
        ctx.save_for_backward(x, w, gamma, y, z if amask is None else None, mean, invstd, amask)
        ctx.beta_ref = beta
        ctx.w_amax = wa          # the weights do not change between this forward and its backward
        ctx.src_bn = getattr(x, "_fs_bn", None)       # x is the output of another conv + BatchNorm layer: its record, for the bwd-data epilogue
        ctx.fan = getattr(x, "_fs_fan", None)         # x is one of the two aliases of a fan-out: this conv may absorb the other alias's gradient
        if ctx.fan is not None:
            FAN_GEOM[ctx.fan[0]] = (tuple(x.shape), tuple(w.shape), meta["stride"], meta["pad"], dil)
        if ctx.fan is not None:
            z._fs_after_fan = ctx.fan[0]              # z is computed FROM the fan-out's conv alias: a layer reading z runs its backward before this one
        rfan = getattr(res, "_fs_fan", None) if res is not None else None
        # The residual gradient may only be left for the fan-out's conv consumer when that consumer's backward is certain to run AFTER this
        # layer's: i.e. when this layer's own conv input was produced by it (BasicBlock: conv1 -> bn1 -> relu -> conv2 + x).  A residual
        # consumer that does not depend on the conv consumer is unordered against it and materialises its gradient as usual.
        ordered = rfan is not None and getattr(x, "_fs_after_fan", None) == rfan[0]
        ctx.res_fan = (rfan[0], FAN_GEOM.get(rfan[0])) if ordered else None
        if act_has_bwd(meta["act"], amask) and meta.get("grad_enabled", True) and any(ctx.needs_input_grad):
            # for the producer of dz (FanOut.backward / a consumer's bwd-data epilogue): this layer's BN-backward operands.  Only when a
            # backward will follow: the record keeps the pre-activation conv output alive as long as z lives
            z._fs_bn = (y, amask, mean, invstd, meta["act"])
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, gamma, y, z, mean, invstd, amask = ctx.saved_tensors
        m = ctx.meta
        dz = dz.contiguous()
        B, Ho, Wo, C = y.shape
        M = B * Ho * Wo
        dy = torch.empty_like(y)
        # residual gradient: absorbed by the bwd-data epilogue of the convolution that shares the fan-out with `res`, where it can be
        absorb = None
        if (m["has_res"] and ctx.res_fan is not None and ctx.res_fan[1] is not None and FUSE_BN_BWD_SUMS and FANOUT
                and (m["act"] == ACT_NONE or amask is not None)):
            xs, wsh, st, pd, dl = ctx.res_fan[1]
            if xs == tuple(y.shape):
                fB, fH, fW, fCin = xs
                fCout, _, fR, fS = wsh
                fHo, fWo = _out_hw(fH, fW, fR, fS, st, pd, dl)
                wsb = hip.conv_workspace_bytes(fH, fW, fCin, fHo, fWo, fCout, fR, fS, st, pd, dl, 1)
                if hip.bwd_data_bnsum_slabs(fB, fH, fW, fCin, fHo, fWo, fCout, fR, fS, st, pd, dl, wsb) > 0 and ctx.res_fan[0] not in FAN_DONE:
                    absorb = ctx.res_fan[0]
        dres = torch.empty_like(y) if (m["has_res"] and absorb is None) else None
        tg, tb = _direct_grad_target(gamma), _direct_grad_target(ctx.beta_ref)
        direct_affine = tg is not None and tb is not None
        dgamma = tg if direct_affine else torch.empty(C, device=y.device, dtype=torch.float32)
        dbeta = tb if direct_affine else torch.empty(C, device=y.device, dtype=torch.float32)
        # BatchNorm backward = column sums (unless the kernel that produced dz already formed them) -> finalize -> apply
        pre = BN_SLABS.pop((dz.data_ptr(), y.data_ptr()), None)
        if pre is not None and pre[2].shape == dz.shape:
            slab, nslab = pre[0], pre[1]
            slab.record_stream(torch.cuda.current_stream())
        else:
            nslab = hip.bn_bwd_slabs(M, C)
            slab = torch.empty(nslab * C * 2, device=y.device, dtype=torch.float32)
            _launch("bn_bwd", 0.0, "fs_bn_bwd_partial", hip.ptr(dz), hip.ptr(z), hip.ptr(amask), hip.ptr(y), hip.ptr(mean), hip.ptr(invstd),
                    M, C, m["act"], hip.ptr(slab))
        coef = torch.empty(4 * C, device=y.device, dtype=torch.float32)
        _launch("bn_bwd", 0.0, "fs_bn_bwd_finalize", hip.ptr(slab), nslab, hip.ptr(gamma), hip.ptr(mean), hip.ptr(invstd), M, C,
                1 if m["training"] else 0, hip.ptr(coef), hip.ptr(dgamma), hip.ptr(dbeta), 1 if direct_affine else 0)
        # timer "work" of the three launches together = algorithmic bytes: dz, y and the mask read once, dy [and dres] written once
        _launch("bn_bwd", 4.0 * M * C * (3 + m["has_res"]) + (M * C // 4 if amask is not None else 0),
                "fs_bn_bwd_apply", hip.ptr(dz), hip.ptr(z), hip.ptr(amask), hip.ptr(y), hip.ptr(coef), M, C, m["act"], float(m["drop_p"]),
                int(m["drop_key"]), hip.ptr(dy), hip.ptr(dres))
        if absorb is not None:
            PENDING_RES[absorb] = (dz, amask if m["act"] != ACT_NONE else None)
        tgt = _direct_grad_target(w)
        # this conv's own input: the other alias's gradient (if a residual consumer stashed it) joins dx in the epilogue, and then dx is
        # the WHOLE gradient of the fan-out's input, so that tensor's producer record (fan tag) is the one the BatchNorm sums are for
        pend = PENDING_RES.pop(ctx.fan[0], None) if ctx.fan is not None else None
        if ctx.fan is not None:
            FAN_DONE.add(ctx.fan[0])          # from here on a residual consumer of this fan-out must materialise its own gradient
        src_bn = ctx.src_bn if ctx.fan is None else (ctx.fan[1] if pend is not None else None)
        if not WGRAD_FIRST:
            dx = conv2d_bwd_data(dy, w, x.shape, m["stride"], m["pad"], m["dil"], w_amax=ctx.w_amax, src_bn=src_bn, addend=pend) if ctx.needs_input_grad[0] else None
        # (not under stream capture: forking the side stream into a capture once per layer and joining it once at the end ends in a host
        #  segmentation fault inside hipStreamEndCapture on ROCm 7.2 -- tools/probes/graph_probe.py, profiles/r05/graph_probe.txt)
        if (tgt is not None and 2.0 * M * C * w.shape[1] * w.shape[2] * w.shape[3] < WGRAD_SIDE_FLOPS and TIMER is None
                and not torch.cuda.is_current_stream_capturing()):
            side = _wgrad_side_stream(dy.device)
            _queue_wgrad_join()                       # backward() returns with its stream ordered behind the side stream
            hip.stream_wait(side, hip._stream())      # dy (and x) were produced on the current stream
            keep = _WGRAD_SIDE_BUSY.setdefault(side, [])
            if len(keep) >= _WGRAD_ROTATE and not torch.cuda.is_current_stream_capturing():
                # a long backward (configs[4]: every layer is small) does not hold on to all of its activations and gradients: lists
                # retire under an event and go when it has completed
                _wgrad_retire(side, keep)
                keep = _WGRAD_SIDE_BUSY[side] = []
                _wgrad_reap()
            keep.append((x, dy))
            hip.set_stream_override(side)
            try:
                conv2d_bwd_weight(x, dy, w.shape, m["stride"], m["pad"], out=tgt, dil=m["dil"], accumulate=True, keep=keep)
            finally:
                hip.set_stream_override(None)
            dw = None
        else:
            dw = conv2d_bwd_weight(x, dy, w.shape, m["stride"], m["pad"], out=tgt, dil=m["dil"], accumulate=tgt is not None)
        if tgt is not None:
            dw = None
        if WGRAD_FIRST:     # dx is what the next backward node reads: produce it last so it is the freshest tensor in the cache
            dx = conv2d_bwd_data(dy, w, x.shape, m["stride"], m["pad"], m["dil"], w_amax=ctx.w_amax, src_bn=src_bn, addend=pend) if ctx.needs_input_grad[0] else None
        dbias = colsum(dy, C) if m["has_bias"] else None
        if direct_affine:
            dgamma = dbeta = None
        return dx, dw, dbias, dgamma, dbeta, dres, None


class ConvBias(Function):
    """y = [dropout_p](conv(x, w) + bias) without normalisation (nn.Linear as a 1x1 conv, patch embeddings, sequence-reduction convs).
    drop_p > 0: the hidden-state Dropout that follows the layer (SegformerSelfOutput / MixFFN) runs in the conv epilogue -- same hash
    of the output's element index as the stand-alone fs_dropout -- and its backward masks dy once before the two gradient launches."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, drop_p=0.0, drop_key=0):
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad, bias is not None, float(drop_p), int(drop_key))
        ctx.bias_ref = bias
        ctx.w_via = getattr(w, "_fs_grad_via", None)      # param_view's record, kept here: the saved tensor may come back as a new object
        return conv2d_fwd(x, w, bias, stride, pad, float(drop_p), int(drop_key))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_bias, drop_p, drop_key = ctx.sp
        dy = dy.contiguous()
        if drop_p > 0.0:
            masked = torch.empty_like(dy)
            hip.call("fs_dropout", hip.ptr(dy), hip.ptr(masked), dy.numel(), drop_p, drop_key)
            dy = masked
        dx = conv2d_bwd_data(dy, w, x.shape, stride, pad) if ctx.needs_input_grad[0] else None
        if ctx.w_via is not None and getattr(w, "_fs_grad_via", None) is None:
            w._fs_grad_via = ctx.w_via
        tgt = _direct_grad_target(w)
        cout, cin, r, s = w.shape
        rows = dy.numel() // cout
        if has_bias and r == 1 and s == 1 and stride == 1 and pad == 0 and hip.linear_bwd_weight_bias_ok(rows, cin, cout):
            # a linear layer: the bias gradient (column sums of dy) rides in the weight-gradient launch, which reads dy anyway
            btgt = _direct_grad_target(ctx.bias_ref)
            dw = rsck(tgt) if tgt is not None else torch.empty(1, 1, cin, cout, device=x.device, dtype=torch.float32)
            db = btgt if btgt is not None else torch.empty(cout, device=x.device, dtype=torch.float32)
            lws_bytes = hip.query("fs_linear_bwd_weight_bias_ws_bytes", cin, cout)      # deterministic mode only
            lws = torch.empty(lws_bytes, device=x.device, dtype=torch.uint8) if lws_bytes else None
            _launch("conv_wgrad", 2.0 * rows * cout * cin, "fs_linear_bwd_weight_bias", hip.ptr(x), hip.ptr(dy), hip.ptr(dw), hip.ptr(db),
                    rows, cin, cout, 1 if tgt is not None else 0, 1 if btgt is not None else 0, hip.ptr(lws), lws_bytes)
            return (dx, None if tgt is not None else dw.permute(3, 2, 0, 1), None if btgt is not None else db, None, None, None, None)
        dw = conv2d_bwd_weight(x, dy, w.shape, stride, pad, out=tgt, accumulate=tgt is not None)
        if tgt is not None:
            dw = None
        db = colsum(dy, dy.shape[-1], into=_direct_grad_target(ctx.bias_ref)) if has_bias else None
        return dx, dw, db, None, None, None, None


# ----------------------------------------------------------------------------------------------
# HRNet fuse + final concat
# ----------------------------------------------------------------------------------------------
class HrFuse(Function):
    """out = relu(sum_t up(t)) over up to 4 NHWC terms (lower-resolution terms bilinearly up-sampled)."""

    @staticmethod
    def forward(ctx, Ho, Wo, *terms):
        n = len(terms)
        B, _, _, C = terms[0].shape
        out = torch.empty(B, Ho, Wo, C, device=terms[0].device, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * n)(*[hip.ptr(t) for t in terms])
        th = (ctypes.c_int * n)(*[t.shape[1] for t in terms])
        tw = (ctypes.c_int * n)(*[t.shape[2] for t in terms])
        hip.call("fs_hr_fuse_fwd", ptrs, th, tw, n, hip.ptr(out), B, Ho, Wo, C, 1)
        ctx.save_for_backward(out)
        ctx.shapes = [tuple(t.shape) for t in terms]
        # terms that are the output of an activation-free conv + BatchNorm layer (the last ConvBn of every fuse path): their output gradient
        # IS this node's gradient (up-sampled back for the lower-resolution ones), so backward forms their BatchNorm-backward sums as it
        # produces that gradient (round 5; before, each of them ran its own reduction pass over it: 62 of the 91 left in an HRNet step)
        ctx.bns = [(bn if (bn is not None and bn[4] == ACT_NONE and bn[1] is None and tuple(bn[0].shape) == tuple(t.shape) and t.shape[-1] <= 1024) else None)
                   for t, bn in ((t, getattr(t, "_fs_bn", None)) for t in terms)] if FUSE_BN_BWD_SUMS else [None] * n
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = dout.contiguous()
        g = torch.empty_like(out)
        B, Ho, Wo, C = out.shape
        same = [i for i, shp in enumerate(ctx.shapes) if ctx.needs_input_grad[2 + i] and shp[1] == Ho and shp[2] == Wo and ctx.bns[i] is not None][:3]
        if same:
            M = B * Ho * Wo
            nslab = hip.bn_bwd_slabs(M, C)
            slabs = [torch.empty(nslab * C * 2, device=out.device, dtype=torch.float32) for _ in same]
            arr = lambda xs: (ctypes.c_void_p * len(xs))(*[hip.ptr(x) for x in xs])      # noqa: E731
            hip.call("fs_relu_bwd_bnsum", hip.ptr(dout), hip.ptr(out), hip.ptr(g), M, C, len(same), arr([ctx.bns[i][0] for i in same]),
                     arr([ctx.bns[i][2] for i in same]), arr([ctx.bns[i][3] for i in same]), arr(slabs))
            for i, slab in zip(same, slabs):
                BN_SLABS[(g.data_ptr(), ctx.bns[i][0].data_ptr())] = (slab, nslab, g)
        else:
            hip.call("fs_relu_bwd", hip.ptr(dout), hip.ptr(out), hip.ptr(g), out.numel())
        grads = []
        for i, shp in enumerate(ctx.shapes):
            if not ctx.needs_input_grad[2 + i]:
                grads.append(None)
            elif shp[1] == Ho and shp[2] == Wo:
                grads.append(g)
            else:
                d = torch.empty(shp, device=out.device, dtype=torch.float32)
                bn = ctx.bns[i]
                if bn is not None and (Ho // shp[1]) % 2 == 0 and (Wo // shp[2]) % 2 == 0:
                    y, _, mean, invstd, _ = bn
                    nslab = hip.bn_bwd_slabs(B * shp[1] * shp[2], C)
                    slab = torch.empty(nslab * C * 2, device=out.device, dtype=torch.float32)
                    hip.call("fs_upsample_slice_bwd_bnsum", hip.ptr(g), B, Ho, Wo, C, 0, hip.ptr(d), shp[1], shp[2], C, hip.ptr(y), hip.ptr(mean),
                             hip.ptr(invstd), hip.ptr(slab))
                    BN_SLABS[(d.data_ptr(), y.data_ptr())] = (slab, nslab, d)
                else:
                    hip.call("fs_upsample_slice_bwd", hip.ptr(g), B, Ho, Wo, C, 0, hip.ptr(d), shp[1], shp[2], C)
                grads.append(d)
        return (None, None, *grads)


class UpsampleTo(Function):
    """F.interpolate(x, size=(Ho,Wo), mode='bilinear', align_corners=False) on NHWC (integer factors)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        B, h, w, C = x.shape
        out = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.float32)
        hip.call("fs_upsample_slice_fwd", hip.ptr(x), B, h, w, C, hip.ptr(out), Ho, Wo, C, 0)
        ctx.shape = (B, h, w, C, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, g):
        B, h, w, C, Ho, Wo = ctx.shape
        d = torch.empty(B, h, w, C, device=g.device, dtype=torch.float32)
        hip.call("fs_upsample_slice_bwd", hip.ptr(g.contiguous()), B, Ho, Wo, C, 0, hip.ptr(d), h, w, C)
        return d, None, None


class UpsampleConcat(Function):
    """cat([x0, up(x1), up(x2), up(x3)], channel) written straight into one NHWC buffer."""

    @staticmethod
    def forward(ctx, *xs):
        B, Ho, Wo, _ = xs[0].shape
        Ct = sum(t.shape[3] for t in xs)
        out = torch.empty(B, Ho, Wo, Ct, device=xs[0].device, dtype=torch.float32)
        off = 0
        for t in xs:
            hip.call("fs_upsample_slice_fwd", hip.ptr(t), B, t.shape[1], t.shape[2], t.shape[3], hip.ptr(out), Ho, Wo, Ct, off)
            off += t.shape[3]
        ctx.shapes = [tuple(t.shape) for t in xs]
        ctx.out_shape = (B, Ho, Wo, Ct)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, Ho, Wo, Ct = ctx.out_shape
        grads, off = [], 0
        for shp in ctx.shapes:
            d = torch.empty(shp, device=g.device, dtype=torch.float32)
            hip.call("fs_upsample_slice_bwd", hip.ptr(g), B, Ho, Wo, Ct, off, hip.ptr(d), shp[1], shp[2], shp[3])
            grads.append(d)
            off += shp[3]
        return tuple(grads)


# ----------------------------------------------------------------------------------------------
# C1 tail
# ----------------------------------------------------------------------------------------------
class MaskHead(Function):
    """m = sigmoid(conv1x1(x) + b) - 0.5; x (B,H,W,C) -> m (B,H,W)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        B, H, W, C = x.shape
        m = torch.empty(B, H, W, device=x.device, dtype=torch.float32)
        hip.call("fs_mask_head_fwd", hip.ptr(x), hip.ptr(w), hip.ptr(bias), hip.ptr(m), B * H * W, C)
        ctx.save_for_backward(x, w, m)
        return m

    @staticmethod
    def backward(ctx, dm):
        x, w, m = ctx.saved_tensors
        dm = dm.contiguous()
        C = x.shape[-1]
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(1, device=x.device, dtype=torch.float32)
        scratch = torch.empty(hip.query("fs_mask_head_bwd_scratch_floats", m.numel(), C), device=x.device, dtype=torch.float32)
        hip.call("fs_mask_head_bwd", hip.ptr(dm), hip.ptr(m), hip.ptr(x), hip.ptr(w), hip.ptr(dx), hip.ptr(dw), hip.ptr(db),
                 m.numel(), C, hip.ptr(scratch))
        return dx, dw, db


class MaxPool(Function):
    """nn.MaxPool2d(k, stride, pad) on NHWC."""

    @staticmethod
    def forward(ctx, x, k, stride, pad):
        B, H, W, C = x.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.float32)
        arg = torch.empty(B, Ho, Wo, C, device=x.device, dtype=torch.int32)
        hip.call("fs_maxpool_fwd", hip.ptr(x), hip.ptr(out), hip.ptr(arg), B, H, W, C, Ho, Wo, k, stride, pad)
        ctx.save_for_backward(arg)
        ctx.cfg = (B, H, W, C, Ho, Wo, k, stride, pad)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, H, W, C, Ho, Wo, k, stride, pad = ctx.cfg
        dx = torch.empty(B, H, W, C, device=g.device, dtype=torch.float32)
        hip.call("fs_maxpool_bwd", hip.ptr(g.contiguous()), hip.ptr(arg), hip.ptr(dx), B, H, W, C, Ho, Wo, k, stride, pad)
        return dx, None, None, None


class Dropout(Function):
    """Stand-alone nn.Dropout(p) with the replayable hash mask (identity when p == 0)."""

    @staticmethod
    def forward(ctx, x, p, key):
        ctx.pk = (float(p), int(key))
        out = torch.empty_like(x)
        hip.call("fs_dropout", hip.ptr(x), hip.ptr(out), x.numel(), float(p), int(key))
        return out

    @staticmethod
    def backward(ctx, g):
        p, key = ctx.pk
        out = torch.empty_like(g)
        hip.call("fs_dropout", hip.ptr(g.contiguous()), hip.ptr(out), g.numel(), p, key)
        return out, None, None


class AvgPoolHW(Function):
    @staticmethod
    def forward(ctx, x):
        B, H, W, C = x.shape
        out = torch.empty(B, C, device=x.device, dtype=torch.float32)
        hip.call("fs_avgpool_fwd", hip.ptr(x), B, H * W, C, hip.ptr(out))
        ctx.shape = (B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, C = ctx.shape
        dx = torch.empty(B, H, W, C, device=g.device, dtype=torch.float32)
        hip.call("fs_avgpool_bwd", hip.ptr(g.contiguous()), B, H * W, C, hip.ptr(dx))
        return dx


class PredAssemble(Function):
    """pred (B,K,H,W): channels < K-1 = class logits broadcast, channel K-1 = logit * mask."""

    @staticmethod
    def forward(ctx, cls, m):
        B, K = cls.shape
        _, H, W = m.shape
        pred = torch.empty(B, K, H, W, device=cls.device, dtype=torch.float32)
        hip.call("fs_pred_assemble_fwd", hip.ptr(cls), hip.ptr(m), hip.ptr(pred), B, K, H * W)
        ctx.save_for_backward(cls, m)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        cls, m = ctx.saved_tensors
        B, K = cls.shape
        dcls = torch.empty_like(cls)
        dm = torch.empty_like(m)
        hip.call("fs_pred_assemble_bwd", hip.ptr(dpred.contiguous()), hip.ptr(cls), hip.ptr(m), hip.ptr(dcls), hip.ptr(dm), B, K,
                 m.shape[1] * m.shape[2])
        return dcls, dm


class SegLoss(Function):
    """(pred (B,K,H,W), gt (B,H,W) int64) -> 7 scalars {dice+focal, focal, dice, 4 accuracies};
    only element 0 carries a gradient."""

    @staticmethod
    def forward(ctx, pred, gt, gamma):
        B, K, H, W = pred.shape
        accum = torch.empty(B * ((H * W + 1023) // 1024) * (3 * K + 7), device=pred.device, dtype=torch.float64)
        out = torch.empty(7, device=pred.device, dtype=torch.float32)
        coef = torch.empty(2 * K, device=pred.device, dtype=torch.float32)
        _launch("fe_seg_loss_fwd", 4.0 * B * H * W * (K + 2), "fs_seg_loss_fwd", hip.ptr(pred), hip.ptr(gt), B, K, H * W, float(gamma), 1e-7, hip.ptr(accum), hip.ptr(out),
                hip.ptr(coef))
        ctx.save_for_backward(pred, gt, coef)
        ctx.gamma = float(gamma)
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, gt, coef = ctx.saved_tensors
        B, K, H, W = pred.shape
        g0 = gout[0:1].contiguous()
        dpred = torch.empty_like(pred)
        _launch("fe_seg_loss_bwd", 4.0 * B * H * W * (2 * K + 2), "fs_seg_loss_bwd", hip.ptr(pred), hip.ptr(gt), hip.ptr(coef), hip.ptr(g0), hip.ptr(dpred), B, K, H * W, ctx.gamma)
        return dpred, None, None


# ----------------------------------------------------------------------------------------------
# foveation front-end
# ----------------------------------------------------------------------------------------------
def gaze_lowres(x, focus, hs, ws):
    B, C, H, W = x.shape
    assert C == 3
    out = torch.empty(B, hs, ws, 5, device=x.device, dtype=torch.float32)
    _launch("fe_gaze_lowres", 4.0 * B * hs * ws * (4 * 3 + 5), "fs_gaze_lowres_fwd", hip.ptr(x), hip.ptr(focus), hip.ptr(out), B, H, W, hs, ws)
    return out


class CompressSoftmax(Function):
    """xs = softmax_HW(conv1x1(relu(s)) + b): s (B,H,W,C) -> xs (B,1,H,W)."""

    @staticmethod
    def forward(ctx, s, w, bias):
        B, H, W, C = s.shape
        xs = torch.empty(B, 1, H, W, device=s.device, dtype=torch.float32)
        hip.call("fs_compress_softmax_fwd", hip.ptr(s), hip.ptr(w), hip.ptr(bias), hip.ptr(xs), B, H * W, C)
        ctx.save_for_backward(s, w, xs)
        return xs

    @staticmethod
    def backward(ctx, g):
        s, w, xs = ctx.saved_tensors
        B, H, W, C = s.shape
        ds = torch.empty_like(s)
        dw = torch.empty_like(w)
        db = torch.empty(1, device=s.device, dtype=torch.float32)
        scratch = torch.empty(hip.query("fs_compress_softmax_bwd_scratch_floats", B, C), device=s.device, dtype=torch.float32)
        hip.call("fs_compress_softmax_bwd", hip.ptr(g.contiguous()), hip.ptr(xs), hip.ptr(s), hip.ptr(w), hip.ptr(ds), hip.ptr(dw),
                 hip.ptr(db), B, H * W, C, hip.ptr(scratch))
        return ds, dw, db


class Compress(Function):
    """CompressNet.forward on its own (models/models.py:360-372): s (B,H,W,C) -> logits (B,1,H,W) = conv1x1(relu(s)) + b."""

    @staticmethod
    def forward(ctx, s, w, bias):
        B, H, W, C = s.shape
        out = torch.empty(B, 1, H, W, device=s.device, dtype=torch.float32)
        hip.call("fs_compress_fwd", hip.ptr(s), hip.ptr(w), hip.ptr(bias), hip.ptr(out), B, H * W, C)
        ctx.save_for_backward(s, w)
        return out

    @staticmethod
    def backward(ctx, g):
        s, w = ctx.saved_tensors
        B, H, W, C = s.shape
        ds = torch.empty_like(s)
        dw = torch.empty_like(w)
        db = torch.empty(1, device=s.device, dtype=torch.float32)
        scratch = torch.empty(B * (C + 1), device=s.device, dtype=torch.float32)
        hip.call("fs_compress_bwd", hip.ptr(g.contiguous()), hip.ptr(s), hip.ptr(w), hip.ptr(ds), hip.ptr(dw), hip.ptr(db), B, H * W, C,
                 hip.ptr(scratch))
        return ds, dw, db


def area_pool(y, hs, ws):
    B, C, H, W = y.shape
    assert C == 1
    out = torch.empty(B, 1, hs, ws, device=y.device, dtype=torch.float32)
    _launch("fe_area_pool", 4.0 * B * (H * W + hs * ws), "fs_area_pool_fwd", hip.ptr(y), hip.ptr(out), B, H, W, hs, ws)
    return out


class EdgeLoss(Function):
    @staticmethod
    def forward(ctx, xs, target, coef):
        loss = torch.empty(1, device=xs.device, dtype=torch.float32)
        # 6 statistics kept for the backward + the per-workgroup partial records of both passes
        stats = torch.empty(hip.query("fs_edge_loss_stats_floats", xs.numel()), device=xs.device, dtype=torch.float32)
        hip.call("fs_edge_loss_fwd", hip.ptr(xs), hip.ptr(target), xs.numel(), float(coef), hip.ptr(loss), hip.ptr(stats))
        ctx.save_for_backward(xs, target, stats)
        ctx.coef = float(coef)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        xs, target, stats = ctx.saved_tensors
        dxs = torch.empty_like(xs)
        hip.call("fs_edge_loss_bwd", hip.ptr(xs), hip.ptr(target), xs.numel(), ctx.coef, hip.ptr(g.reshape(1).contiguous()),
                 hip.ptr(stats), hip.ptr(dxs))
        return dxs, None, None


PAD_MODES = {"replication": 0, "reflect": 1, "zero": 2}      # TRAIN.def_saliency_pad_mode (models/models.py:819-825)


class GaussGrid(Function):
    """xs (B,1,hs,ws) -> grid (B,hs,ws,2); g1d = separable Gaussian taps (float64, device); pad_mode = PAD_MODES[...] (the padded map
    is never written: the kernels read the source pixel a padded position copies)."""

    @staticmethod
    def forward(ctx, xs, g1d, pad, pad_mode=0):
        B, _, hs, ws = xs.shape
        grid = torch.empty(B, hs, ws, 2, device=xs.device, dtype=torch.float32)
        if pad_mode == 0:
            _launch("fe_gauss_grid_fwd", 4.0 * B * hs * ws * 3, "fs_gauss_grid_fwd", hip.ptr(xs), hip.ptr(g1d), hip.ptr(grid), B, hs, ws, pad)
        else:
            _launch("fe_gauss_grid_fwd", 4.0 * B * hs * ws * 3, "fs_gauss_grid_fwd_mode", hip.ptr(xs), hip.ptr(g1d), hip.ptr(grid), B, hs, ws, pad,
                    pad_mode)
        ctx.save_for_backward(xs, g1d)
        ctx.pad, ctx.pad_mode = pad, pad_mode
        return grid

    @staticmethod
    def backward(ctx, dgrid):
        xs, g1d = ctx.saved_tensors
        B, _, hs, ws = xs.shape
        dxs = torch.empty_like(xs)
        scratch = torch.empty(hip.query("fs_gauss_grid_bwd_scratch_floats", B, hs, ws), device=xs.device, dtype=torch.float32)
        if ctx.pad_mode == 0:
            _launch("fe_gauss_grid_bwd", 4.0 * B * hs * ws * 4, "fs_gauss_grid_bwd", hip.ptr(xs), hip.ptr(g1d), hip.ptr(dgrid.contiguous()), hip.ptr(dxs),
                    B, hs, ws, ctx.pad, hip.ptr(scratch))
        else:
            _launch("fe_gauss_grid_bwd", 4.0 * B * hs * ws * 4, "fs_gauss_grid_bwd_mode", hip.ptr(xs), hip.ptr(g1d), hip.ptr(dgrid.contiguous()),
                    hip.ptr(dxs), B, hs, ws, ctx.pad, ctx.pad_mode, hip.ptr(scratch))
        return dxs, None, None, None


class GridUpsample(Function):
    """nn.Upsample(size=(H,W), mode='bilinear') of the grid (B,h,w,2) -> (B,H,W,2) (models/models.py:621-631)."""

    @staticmethod
    def forward(ctx, grid, H, W):
        B, h, w, _ = grid.shape
        out = torch.empty(B, H, W, 2, device=grid.device, dtype=torch.float32)
        hip.call("fs_grid_upsample_fwd", hip.ptr(grid.contiguous()), hip.ptr(out), B, h, w, H, W)
        ctx.cfg = (B, h, w, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, h, w, H, W = ctx.cfg
        d = torch.empty(B, h, w, 2, device=g.device, dtype=torch.float32)
        hip.call("fs_grid_upsample_bwd", hip.ptr(g.contiguous()), hip.ptr(d), B, h, w, H, W)
        return d, None, None


class GridSample(Function):
    """x (B,C,H,W) NCHW, grid (B,h,w,2) -> (B,h,w,C) NHWC; gradient w.r.t. the grid (and, when asked,
    w.r.t. x by scatter-add)."""

    @staticmethod
    def forward(ctx, x, grid):
        B, C, H, W = x.shape
        _, h, w, _ = grid.shape
        out = torch.empty(B, h, w, C, device=x.device, dtype=torch.float32)
        _launch("fe_grid_sample_fwd", 4.0 * B * h * w * (4 * C + 2 + C), "fs_grid_sample_fwd", hip.ptr(x), hip.ptr(grid), hip.ptr(out), B, C, H, W, h, w, 1)
        ctx.save_for_backward(x, grid)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, grid = ctx.saved_tensors
        B, C, H, W = x.shape
        _, h, w, _ = grid.shape
        gout = gout.contiguous()
        dx = dgrid = None
        if ctx.needs_input_grad[1]:
            dgrid = torch.empty_like(grid)
            _launch("fe_grid_sample_bwd_grid", 4.0 * B * h * w * (4 * C + C + 2 + 2), "fs_grid_sample_bwd_grid", hip.ptr(gout), hip.ptr(x), hip.ptr(grid), hip.ptr(dgrid), B, C, H, W, h, w, 1)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            hip.call("fs_grid_sample_bwd_input", hip.ptr(gout), hip.ptr(grid), hip.ptr(dx), B, C, H, W, h, w, 1)
        return dx, dgrid


def grid_sample_label(y, grid, return_float=False):
    B, C, H, W = y.shape
    assert C == 1
    _, h, w, _ = grid.shape
    label = torch.empty(B, h, w, device=y.device, dtype=torch.int64)
    ys = torch.empty(B, h, w, device=y.device, dtype=torch.float32) if return_float else None
    _launch("fe_grid_sample_label", 4.0 * B * h * w * (4 + 2 + 2), "fs_grid_sample_label", hip.ptr(y), hip.ptr(grid), hip.ptr(label), hip.ptr(ys), B, H, W, h, w)
    return (label, ys) if return_float else label


def inverse_index_maps(grid, H, W):
    n = grid.numel() // 2
    u = torch.empty(grid.shape[:-1], device=grid.device, dtype=torch.int64)
    v = torch.empty_like(u)
    hip.call("fs_inverse_index_maps", hip.ptr(grid), hip.ptr(u), hip.ptr(v), n, H, W)
    return u, v


def inverse_grid(grid, Hs, Ws):
    """models/models.py:639-655: (owner (B,Hs,Ws) int32 with -1 in holes, grid_inv (B,Hs,Ws,2) with 0 in holes)."""
    B, h, w, _ = grid.shape
    owner = torch.empty(B, Hs, Ws, device=grid.device, dtype=torch.int32)
    inv = torch.empty(B, Hs, Ws, 2, device=grid.device, dtype=torch.float32)
    hip.call("fs_inverse_grid", hip.ptr(grid.contiguous()), hip.ptr(owner), hip.ptr(inv), B, h, w, Hs, Ws)
    return owner, inv


def unwarp_nearest(pred, grid, Hs, Ws):
    """Full-resolution prediction from the foveated one (no autograd): pred (B,C,h,w) is sampled through the inverse grid
    (F.grid_sample(pred, grid_inv), models/models.py:933) and the never-claimed pixels take their nearest claimed
    neighbour (rev_deform_interp='nearest').  Returns (pred_full (B,C,Hs,Ws), hole mask (B,Hs,Ws) bool)."""
    B, C, h, w = pred.shape
    owner, inv = inverse_grid(grid, Hs, Ws)
    out = torch.empty(B, C, Hs, Ws, device=pred.device, dtype=torch.float32)
    hip.call("fs_grid_sample_fwd", hip.ptr(pred.contiguous()), hip.ptr(inv), hip.ptr(out), B, C, h, w, Hs, Ws, 0)
    scratch = torch.empty(2 * B * Hs * Ws, device=pred.device, dtype=torch.int32)
    hip.call("fs_fill_nearest", hip.ptr(out), hip.ptr(owner), hip.ptr(scratch), B, C, Hs, Ws)
    return out, owner < 0


# ----------------------------------------------------------------------------------------------
# SegFormer pieces (tokens = NHWC rows)
# ----------------------------------------------------------------------------------------------
class LayerNorm(Function):
    """nn.LayerNorm over the last dimension."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        hip.call("fs_layernorm_fwd", hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd), M, C, float(eps))
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta_ref = beta
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        C = x.shape[-1]
        dx = torch.empty_like(x)
        tg, tb = _direct_grad_target(gamma), _direct_grad_target(ctx.beta_ref)
        direct = tg is not None and tb is not None
        dgamma = tg if direct else torch.empty_like(gamma)
        dbeta = tb if direct else torch.empty_like(gamma)
        M = x.numel() // C
        scratch = torch.empty(hip.query("fs_layernorm_bwd_scratch_floats", M, C), device=x.device, dtype=torch.float32)
        hip.call("fs_layernorm_bwd", hip.ptr(g.contiguous()), hip.ptr(x), hip.ptr(gamma), hip.ptr(mean), hip.ptr(rstd), hip.ptr(dx),
                 hip.ptr(dgamma), hip.ptr(dbeta), M, C, 1 if direct else 0, hip.ptr(scratch))
        if direct:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None


class LayerNormFan(Function):
    """(LayerNorm(x), x): the pre-norm fan-out of a transformer block -- x feeds the normalisation AND the residual add -- as one autograd
    node, so that both gradients of x arrive together and the LayerNorm backward pass adds the residual's while it writes dx
    (fs_layernorm_bwd_add) instead of the autograd engine running an add pass over the activation (104 per configs[3] step)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        hip.call("fs_layernorm_fwd", hip.ptr(x), hip.ptr(gamma), hip.ptr(beta), hip.ptr(y), hip.ptr(mean), hip.ptr(rstd), M, C, float(eps))
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta_ref = beta
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None, not as a tensor of zeros
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_skip):
        x, gamma, mean, rstd = ctx.saved_tensors
        if g is None:
            return g_skip, None, None, None
        C = x.shape[-1]
        dx = torch.empty_like(x)
        tg, tb = _direct_grad_target(gamma), _direct_grad_target(ctx.beta_ref)
        direct = tg is not None and tb is not None
        dgamma = tg if direct else torch.empty_like(gamma)
        dbeta = tb if direct else torch.empty_like(gamma)
        M = x.numel() // C
        scratch = torch.empty(hip.query("fs_layernorm_bwd_scratch_floats", M, C), device=x.device, dtype=torch.float32)
        hip.call("fs_layernorm_bwd_add", hip.ptr(g.contiguous()), hip.ptr(x), hip.ptr(gamma), hip.ptr(mean), hip.ptr(rstd),
                 hip.ptr(g_skip.contiguous()) if g_skip is not None else None, hip.ptr(dx), hip.ptr(dgamma), hip.ptr(dbeta), M, C,
                 1 if direct else 0, hip.ptr(scratch))
        if direct:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None


class Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        hip.call("fs_gelu_fwd", hip.ptr(x), hip.ptr(y), x.numel())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        hip.call("fs_gelu_bwd", hip.ptr(g.contiguous()), hip.ptr(x), hip.ptr(dx), x.numel())
        return dx


class Unfold(Function):
    """(B,H,W,C) -> (1, B*Ho*Wo, 1, Kp): k x k patches as rows, element order (r, s, c), zero-padded to Kp columns; backward folds."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, kp):
        B, H, W, C = x.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        col = torch.empty(1, B * Ho * Wo, 1, kp, device=x.device, dtype=torch.float32)
        hip.call("fs_unfold", hip.ptr(x), hip.ptr(col), B, H, W, C, k, stride, pad, Ho, Wo, kp)
        ctx.geom = (B, H, W, C, k, stride, pad, Ho, Wo, kp)
        return col

    @staticmethod
    def backward(ctx, g):
        B, H, W, C, k, stride, pad, Ho, Wo, kp = ctx.geom
        dx = torch.empty(B, H, W, C, device=g.device, dtype=torch.float32)
        hip.call("fs_fold", hip.ptr(g.contiguous()), hip.ptr(dx), B, H, W, C, k, stride, pad, Ho, Wo, kp)
        return dx, None, None, None, None


UNFOLD_BIG_FILTERS = os.environ.get("FS_UNFOLD_CONV", "1") != "0"
PATCHIFY_DIRECT = os.environ.get("FS_PATCHIFY_DIRECT", "0") == "1"


def conv_bias_any(x, w, bias, stride, pad):
    """ConvBias, with filters of more than 32 taps (which the aligned split-precision kernels do not take: they would run on the generic fp32
    kernels) unfolded into patch rows and computed as one linear layer.  The weight's RSCK storage [r][s][c][k] IS the (k*k*C, Cout) matrix
    of that layer: a view, no copy, and its gradient lands in the same storage."""
    cout, cin, r, s = w.shape
    # ... and so are filters whose stride equals their size (the 2x2 / 4x4 sequence-reduction convs of the Mix-Transformer): their patches
    # do not overlap, the unfold is a pure permutation, and the alternative is one single-tap launch per filter tap and pass (configs[3]:
    # 312 conv_igemm_split launches per step, 13.7 ms; round 4)
    patchify = r > 1 and stride == r and pad == 0
    # (FS_PATCHIFY_DIRECT=1: leave the <= 9-tap ones -- the 2x2 reduction of the Mix-Transformer's 40-block stage -- to the library's own
    # stride >= filter route instead: gathered-row GEMM forward, one scattered-row GEMM per tap for bwd-data, bwd-weight on gathered rows;
    # no unfold / fold passes, no patch matrix kept.  Measured on configs[3], B = 16, alternating on one box: 198.9 / 198.4 ms per step
    # unfolded, 200.4 / 200.4 direct (-0.9 %, 1.2 GB less memory): four quarter-size GEMM launches cost more than one GEMM + a 73 us fold.  Off.)
    if patchify and PATCHIFY_DIRECT and r * s <= 9 and cin % 64 == 0 and cout % 64 == 0:
        patchify = False
    if (not UNFOLD_BIG_FILTERS or r != s or (r * s <= 32 and not patchify) or hip.get_conv_precision() != "bf16x3" or cout % 4 or cout < 16):
        return ConvBias.apply(x, w, bias, stride, pad)
    B, H, W, _ = x.shape
    Ho, Wo = _out_hw(H, W, r, s, stride, pad)
    kk = r * s * cin
    kp = max(16, (kk + 3) // 4 * 4)
    as_matrix = lambda t: t.permute(2, 3, 1, 0).reshape(1, 1, kk, cout).permute(3, 2, 0, 1)      # logical (Cout, kk, 1, 1) over the same RSCK storage
    w2 = param_view(w, as_matrix) if kp == kk else as_matrix(w)
    if kp != kk:
        w2 = PadWeightChannels.apply(w2, kp)
    y = ConvBias.apply(Unfold.apply(x, r, stride, pad, kp), w2, bias, 1, 0)
    return y.view(B, Ho, Wo, cout)


class GeluDropout(Function):
    """dropout_p(gelu(x)) in one pass each way (MixFFN's activation + hidden dropout); the mask is Dropout's for the same key."""

    @staticmethod
    def forward(ctx, x, p, key):
        y = torch.empty_like(x)
        hip.call("fs_gelu_dropout_fwd", hip.ptr(x), hip.ptr(y), x.numel(), float(p), int(key))
        ctx.save_for_backward(x)
        ctx.pk = (float(p), int(key))
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        p, key = ctx.pk
        dx = torch.empty_like(x)
        hip.call("fs_gelu_dropout_bwd", hip.ptr(g.contiguous()), hip.ptr(x), hip.ptr(dx), x.numel(), p, key)
        return dx, None, None


DWCONV_BIAS_FUSED = os.environ.get("FS_DWCONV_BIAS_FUSED", "1") != "0"      # A/B switch: 0 = separate column-sum pass


class DwConv3(Function):
    """Depthwise Conv2d(C,C,3,1,1,groups=C) + bias on NHWC; w logical (C,1,3,3)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        B, H, W, C = x.shape
        y = torch.empty_like(x)
        hip.call("fs_dwconv3_fwd", hip.ptr(x), hip.ptr(w), hip.ptr(bias), hip.ptr(y), B, H, W, C, 0)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, H, W, C = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        hip.call("fs_dwconv3_fwd", hip.ptr(g), hip.ptr(w), None, hip.ptr(dx), B, H, W, C, 1)
        tgt = _direct_grad_target(w)
        tgt = tgt if (tgt is not None and tgt.is_contiguous()) else None
        dw = tgt if tgt is not None else torch.empty_like(w)
        lanes = hip.query("fs_dwconv3_wgrad_lanes", B, H, W, C)
        if ctx.has_bias and not DWCONV_BIAS_FUSED:
            ws = torch.empty(lanes * 9 * C, device=x.device, dtype=torch.float32)
            hip.call("fs_dwconv3_bwd_weight", hip.ptr(x), hip.ptr(g), hip.ptr(dw), hip.ptr(ws), B, H, W, C, 1 if tgt is not None else 0)
            return dx, (None if tgt is not None else dw), colsum(g, C, into=_direct_grad_target(ctx.bias_ref))
        if ctx.has_bias:          # the bias gradient rides in the weight-gradient launches, which read dy once anyway
            btgt = _direct_grad_target(ctx.bias_ref)
            db = btgt if btgt is not None else torch.empty(C, device=x.device, dtype=torch.float32)
            ws = torch.empty(lanes * 10 * C, device=x.device, dtype=torch.float32)
            hip.call("fs_dwconv3_bwd_weight_bias", hip.ptr(x), hip.ptr(g), hip.ptr(dw), hip.ptr(db), hip.ptr(ws), B, H, W, C,
                     1 if tgt is not None else 0, 1 if btgt is not None else 0)
            return dx, (None if tgt is not None else dw), (None if btgt is not None else db)
        ws = torch.empty(lanes * 9 * C, device=x.device, dtype=torch.float32)
        hip.call("fs_dwconv3_bwd_weight", hip.ptr(x), hip.ptr(g), hip.ptr(dw), hip.ptr(ws), B, H, W, C, 1 if tgt is not None else 0)
        if tgt is not None:
            dw = None
        return dx, dw, None


LINEAR_RESIDUAL = os.environ.get("FS_LINEAR_RESIDUAL", "1") != "0"      # A/B switch: 0 = linear layer, then a residual + DropPath pass


class LinearResidual(Function):
    """out = res + DropPath_p2(Dropout_p(x W^T + b)): the linear layer that ends a residual branch of a transformer block with the residual add
    in the GEMM epilogue (fs_conv2d_fwd_residual), and ONE mask pass over the gradient in the backward (fs_droppath_dropout_bwd) where the
    separate nodes ran two (DropPath's, then the Dropout's).  x (..., Cin), res (B, ..., Cout); samples = res.shape[0]."""

    @staticmethod
    def forward(ctx, x, w, bias, res, drop_p, drop_key, dp_p, dp_key):
        cout, cin = w.shape[0], w.shape[1]
        rows = x.numel() // cin
        x4 = x.reshape(1, rows, 1, cin)
        rps = rows // res.shape[0]
        ws, ws_bytes, packed = _pack_for(w, x.device, 1, rows, 1, cin, rows, 1, cout, 1, 1, 1, 0, 1, 0)
        out = torch.empty_like(res)
        _launch_conv(packed, "conv_affine", 2.0 * rows * cout * cin, "fs_conv2d_fwd_residual", hip.ptr(x4), hip.ptr(rsck(w)), hip.ptr(bias),
                     hip.ptr(res), hip.ptr(out), 1, rows, 1, cin, rows, 1, cout, 1, 1, 1, 0, 1, float(drop_p), int(drop_key), float(dp_p), int(dp_key),
                     rps, hip.ptr(ws), ws_bytes, None)
        ctx.save_for_backward(x4, w)
        ctx.cfg = (float(drop_p), int(drop_key), float(dp_p), int(dp_key), rps, bias is not None, tuple(x.shape))
        ctx.bias_ref = bias
        ctx.w_via = getattr(w, "_fs_grad_via", None)
        return out

    @staticmethod
    def backward(ctx, g):
        x4, w = ctx.saved_tensors
        drop_p, drop_key, dp_p, dp_key, rps, has_bias, x_shape = ctx.cfg
        cout, cin = w.shape[0], w.shape[1]
        g = g.contiguous()
        rows = x4.shape[1]
        if drop_p > 0.0 or dp_p > 0.0:
            dy = torch.empty_like(g)
            hip.call("fs_droppath_dropout_bwd", hip.ptr(g), hip.ptr(dy), g.numel(), rps * cout, dp_p, dp_key, drop_p, drop_key)
        else:
            dy = g
        dy4 = dy.view(1, rows, 1, cout)
        dx = conv2d_bwd_data(dy4, w, x4.shape, 1, 0).view(x_shape) if ctx.needs_input_grad[0] else None
        if ctx.w_via is not None and getattr(w, "_fs_grad_via", None) is None:
            w._fs_grad_via = ctx.w_via
        dw, db = _linear_param_grads(x4, dy4, w, ctx.bias_ref if has_bias else None)
        return dx, dw, db, g, None, None, None, None


def _linear_param_grads(x4, dy4, w, bias):
    """(dw, db) of a 1x1 / linear layer as autograd wants them: None where the kernel added straight into the parameter's arena slice."""
    tgt = _direct_grad_target(w)
    cout, cin = w.shape[0], w.shape[1]
    rows = dy4.numel() // cout
    if bias is not None and hip.linear_bwd_weight_bias_ok(rows, cin, cout):
        btgt = _direct_grad_target(bias)
        dw = rsck(tgt) if tgt is not None else torch.empty(1, 1, cin, cout, device=x4.device, dtype=torch.float32)
        db = btgt if btgt is not None else torch.empty(cout, device=x4.device, dtype=torch.float32)
        lws_bytes = hip.query("fs_linear_bwd_weight_bias_ws_bytes", cin, cout)      # deterministic mode only
        lws = torch.empty(lws_bytes, device=x4.device, dtype=torch.uint8) if lws_bytes else None
        _launch("conv_wgrad", 2.0 * rows * cout * cin, "fs_linear_bwd_weight_bias", hip.ptr(x4), hip.ptr(dy4), hip.ptr(dw), hip.ptr(db),
                rows, cin, cout, 1 if tgt is not None else 0, 1 if btgt is not None else 0, hip.ptr(lws), lws_bytes)
        return (None if tgt is not None else dw.permute(3, 2, 0, 1)), (None if btgt is not None else db)
    dw = conv2d_bwd_weight(x4, dy4, w.shape, 1, 0, out=tgt, accumulate=tgt is not None)
    db = colsum(dy4, cout, into=_direct_grad_target(bias)) if bias is not None else None
    return (None if tgt is not None else dw), db


def linear_residual(x, w, bias, res, drop_p, drop_key, dp_p, dp_key):
    """res + DropPath(Dropout(linear(x))): fused where the 1x1 GEMM kernel runs the layer (bf16x3 / f16x2, >= 128 rows per sample), else None
    (the caller composes ConvBias + ResidualDropPath)."""
    if not (LINEAR_RESIDUAL and x.is_cuda and hip.get_conv_precision() != "f32"):
        return None
    cout, cin = w.shape[0], w.shape[1]
    rows = x.numel() // cin
    if rows % res.shape[0] or rows * max(cin, cout) * 4 >= MAX_TENSOR_BYTES:
        return None
    wsb = hip.conv_workspace_bytes(rows, 1, cin, rows, 1, cout, 1, 1, 1, 0, 1, 0)
    if not hip.fwd_residual_ok(1, rows, 1, cin, rows, 1, cout, 1, 1, 1, 0, 1, rows // res.shape[0], wsb):
        return None
    return LinearResidual.apply(x, w, bias, res.contiguous(), drop_p, drop_key, dp_p, dp_key)


class ResidualDropPath(Function):
    """out = x + DropPath_p(y): per-sample keep from the hash (p = 0: plain residual add)."""

    @staticmethod
    def forward(ctx, x, y, p, key):
        out = torch.empty_like(x)
        per = x.numel() // x.shape[0]
        hip.call("fs_residual_droppath", hip.ptr(x), hip.ptr(y), hip.ptr(out), x.numel(), per, float(p), int(key))
        ctx.cfg = (float(p), int(key), per)
        return out

    @staticmethod
    def backward(ctx, g):
        p, key, per = ctx.cfg
        g = g.contiguous()
        if p == 0.0:
            return g, g, None, None
        dy = torch.empty_like(g)
        hip.call("fs_residual_droppath", None, hip.ptr(g), hip.ptr(dy), g.numel(), per, p, key)
        return g, dy, None, None


ATTN_SPLIT = os.environ.get("FS_ATTN_SPLIT", "1") != "0"      # kernel A/B: 0 keeps the exact-fp32 MFMA attention kernels in every mode


class Attention(Function):
    """softmax(q k^T / sqrt(64)) (dropout p) v per head on the matrix cores; q (B,N,C), k/v (B,Nk,C), C = heads*64, any Nk."""

    @staticmethod
    def forward(ctx, q, k, v, heads, p, key, grad_enabled=None):
        # grad_enabled: the grad mode of the CALL SITE (ops.attention records it) -- inside Function.forward grad mode is always off, and
        # ctx.needs_input_grad mirrors requires_grad of the inputs even under no_grad; None (a direct .apply) = decide on needs_input_grad
        will_backward = any(ctx.needs_input_grad[:3]) and (grad_enabled is None or grad_enabled)
        B, N, C = q.shape
        Nk = k.shape[1]
        assert C == heads * 64, "head_dim must be 64"
        o = torch.empty_like(q)
        lse = torch.empty(B * heads * N, device=q.device, dtype=torch.float32)
        mask = None
        split = ATTN_SPLIT and hip.get_conv_precision() == "bf16x3"
        if split:
            # the headline arithmetic (24-bit operands as three bf16 planes, fp32 accumulation) on the attention products as well
            nb = hip.attention_split_ws_bytes(B, Nk, heads)
            ws = torch.empty(nb, device=q.device, dtype=torch.uint8)
            if p > 0 and will_backward:
                # one keep bit per (query, key), left by the forward so that the three backward kernels do not hash every element again
                mask = torch.empty(hip.query("fs_attention_mask_words", B, N, Nk, heads), device=q.device, dtype=torch.int32)
            _launch("attn_fwd", 4.0 * B * heads * N * Nk * 64, "fs_attention_fwd_split", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(o),
                    hip.ptr(lse), hip.ptr(mask) if mask is not None else None, hip.ptr(ws), nb, B, N, Nk, heads, 0.125, float(p), int(key))
        else:
            _launch("attn_fwd", 4.0 * B * heads * N * Nk * 64, "fs_attention_fwd", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(o), hip.ptr(lse),
                    B, N, Nk, heads, 0.125, float(p), int(key))
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.cfg = (heads, float(p), int(key))
        ctx.keep_mask = mask
        ctx.split = split          # the backward takes the kernel family (and workspace contract) the forward prepared for, whatever the mode is by then
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, o, lse = ctx.saved_tensors
        heads, p, key = ctx.cfg
        B, N, C = q.shape
        Nk = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        scratch = torch.empty(B * heads * N, device=q.device, dtype=torch.float32)
        if ctx.split:
            nb = hip.attention_split_ws_bytes(B, Nk, heads, backward=True)
            ws = torch.empty(nb, device=q.device, dtype=torch.uint8)
            mask = ctx.keep_mask
            _launch("attn_bwd", 14.0 * B * heads * N * Nk * 64, "fs_attention_bwd_split", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(o),
                    hip.ptr(go.contiguous()), hip.ptr(lse), hip.ptr(mask) if mask is not None else None, hip.ptr(dq), hip.ptr(dk), hip.ptr(dv),
                    hip.ptr(scratch), hip.ptr(ws), nb, B, N, Nk, heads, 0.125, p, key)
        else:
            _launch("attn_bwd", 14.0 * B * heads * N * Nk * 64, "fs_attention_bwd", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(o),
                    hip.ptr(go.contiguous()), hip.ptr(lse), hip.ptr(dq), hip.ptr(dk), hip.ptr(dv), hip.ptr(scratch), B, N, Nk, heads, 0.125, p, key)
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, heads, p, key):
    """Attention.apply with the call site's grad mode (decides whether the forward leaves the dropout keep words for the backward)."""
    return Attention.apply(q, k, v, heads, p, key, torch.is_grad_enabled())
