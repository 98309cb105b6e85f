"""Input-pipeline step in front of the path.

The reference's `PreprocessDataset.__getitem__` (DynamicFocus/e_preprocess_scripts/dataset.py:127-142) turns a decoded
LVIS sample into float tensors on the host -- RGBA float32 (4,HP,WP) + float mask, 12.6-50 MB per image -- and the
training loop ships them to the GPU.  Here the sample crosses PCIe as it was decoded (uint8 HWC image + uint8 mask, a
quarter of the bytes, from pinned staging buffers on a copy stream) and `fs_ingest_sample` does ToTensor (/255), the
zero padding and the batch assembly on the device, bit-identically.  File-name / sample contract unchanged:
`(X_4xHPxWP, F_2, Y_1xHPxWP, Y_cls_1)` per sample, default-collated to `(B,4,HP,WP), (B,2), (B,1,HP,WP), (B,1)`.
"""
import numpy as np
import torch

from . import hip


def _as_u8(a):
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a.contiguous()
    if t.dtype != torch.uint8:
        raise TypeError(f"decoded samples travel as uint8, got {t.dtype}")
    return t


class Sample:
    """One decoded sample: img (H,W,Ci) uint8 in PIL memory order, mask (H,W) uint8, pads = (left, right, top, bottom),
    focus = (idx_H, idx_W) in the padded frame, frame = (HC, WC) the focus is normalised by, cls = class id."""

    def __init__(self, img, mask, pads, focus, frame, cls):
        self.img, self.mask = _as_u8(img), _as_u8(mask)
        assert self.img.dim() == 3 and self.mask.shape == self.img.shape[:2], (self.img.shape, self.mask.shape)
        self.pads = tuple(int(p) for p in pads)
        self.focus, self.frame, self.cls = focus, frame, int(cls)

    @property
    def padded_hw(self):
        H, W = self.mask.shape
        l, r, t, b = self.pads
        return H + t + b, W + l + r


def ingest_batch(samples, device, channels=4, staged=None):
    """Device batch (X, F, Y, cls) of `samples`; all samples must pad to the same (HP, WP).
    staged: optional list of (img_dev, mask_dev) uint8 device tensors already copied (DevicePrefetcher)."""
    HP, WP = samples[0].padded_hw
    B = len(samples)
    X = torch.empty(B, channels, HP, WP, device=device, dtype=torch.float32)
    Y = torch.empty(B, 1, HP, WP, device=device, dtype=torch.float32)
    for b, s in enumerate(samples):
        if s.padded_hw != (HP, WP):
            raise ValueError(f"sample {b} pads to {s.padded_hw}, batch is {(HP, WP)}")
        img, mask = staged[b] if staged is not None else (s.img.to(device, non_blocking=True), s.mask.to(device, non_blocking=True))
        H, W, Ci = s.img.shape
        l, r, t, bt = s.pads
        hip.call("fs_ingest_sample", hip.ptr(img), hip.ptr(mask), hip.ptr(X), hip.ptr(Y), b, H, W, Ci, channels, l, r, t, bt)
    F = torch.tensor([[s.focus[0] / s.frame[0], s.focus[1] / s.frame[1]] for s in samples], dtype=torch.float32).to(device, non_blocking=True)
    cls = torch.tensor([[s.cls] for s in samples], dtype=torch.int64).to(device, non_blocking=True)
    return X, F, Y, cls


class DevicePrefetcher:
    """Iterates device batches one batch ahead of the consumer: pinned uint8 staging -> async H2D on a copy stream ->
    `fs_ingest_sample` on that stream; the consumer's stream waits on the batch's event, so the copy and the conversion of
    batch k+1 overlap the training step of batch k."""

    def __init__(self, batches, device, channels=4):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.channels = channels
        self.stream = torch.cuda.Stream(device=self.device)
        self._next = None
        self._preload()

    def _preload(self):
        try:
            samples = next(self.it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            staged = []
            for s in samples:
                img = s.img.pin_memory().to(self.device, non_blocking=True)
                mask = s.mask.pin_memory().to(self.device, non_blocking=True)
                staged.append((img, mask))
            batch = ingest_batch(samples, self.device, self.channels, staged=staged)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._next = (batch, ev, staged)

    def __iter__(self):
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        batch, ev, staged = self._next
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for t in batch:
            t.record_stream(cur)          # allocated on the copy stream, consumed on the caller's
        del staged
        self._preload()
        return batch
