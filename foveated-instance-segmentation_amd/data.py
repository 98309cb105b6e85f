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


# ----------------------------------------------------------------------------------------------------------------
# the reference's dataset directory / file-name contract (DynamicFocus/e_preprocess_scripts/dataset.py:44-125)
# ----------------------------------------------------------------------------------------------------------------
# cityscapes class remapping of the reference (dataset.py:13-36): ids outside the table map to 19
CITYSCAPES_IDX = {6: 0, 2: 1, 17: 2, 12: 3, 13: 4, 10: 5, 4: 6, 18: 7, 26: 8, 22: 9, 32: 10, 0: 11, 19: 12, 37: 13, 28: 14, 8: 15,
                  31: 16, 25: 17, 30: 18}


class PreprocessDataset:
    """`PreprocessDataset(data_path, marker, dataset_partition, dataset_name)` of the reference, same constructor, same
    `data_info` records, same `len()`; `__getitem__` returns the DECODED sample (`Sample`: uint8 image + uint8 mask + pads + focus
    + class) instead of float tensors -- ToTensor, zero padding and batch assembly happen on the device (`ingest_batch`,
    `DevicePrefetcher`), bit-identically to the reference's `__getitem__` + default collate.

    Layout read: `<data_path>/<dataset_name>/<partition>/<marker>/*.Y.pt`.
      lvis       `caty_cid_kid_aid_imgid_fpos_paddings_IxHxW.uint8.Y.pt`: class id = int(kid[1:]), focus = fpos "HxW", paddings
                 "LxRxTxB"; the image is `<imgid>.jpg` under the partition's COCO directory, else the first of train / val / test
                 2017 that has it (dataset.py:93-101).  `coco_root` = the `coco2017` directory (the reference takes it from a
                 host-name table in `preset.py`; here an argument or $FS_COCO_ROOT).
      cityscapes `caty_cid_kid_itemkey_fpos_IxHxW.uint8.Y.pt` + the matching `..._3xHxW.uint8.X.pt` uint8 image tensor; class id
                 through the reference's 19-class table.
    HC = WC = 640 for every dataset name: the reference compares against the misspelt 'cityscpaes' (dataset.py:46-47), so its
    512 x 1024 branch is never taken; the focus is normalised by 640 in both directions, as there."""

    def __init__(self, data_path, marker, dataset_partition="train", dataset_name="cityscapes", transform=None, coco_root=None):
        import os
        self.HC = 512 if dataset_name == "cityscpaes" else 640
        self.WC = 1024 if dataset_name == "cityscpaes" else 640
        self.K = len(CITYSCAPES_IDX)
        self.data_path, self.marker, self.dataset_partition, self.dataset_name = data_path, marker, dataset_partition, dataset_name
        if transform is not None:
            raise NotImplementedError("the reference's training script passes no transform; host-side float transforms are not part of this pipeline")
        coco_root = coco_root or os.environ.get("FS_COCO_ROOT", "")
        coco = {p: os.path.join(coco_root, p) for p in ("train2017", "val2017", "test2017")}
        self.coco_path = coco["train2017"] if dataset_partition == "train" else coco["val2017"]
        self.dpath_data_cook_data_part_mark = os.path.join(data_path, dataset_name, dataset_partition, marker)
        self.data_info = []
        root = self.dpath_data_cook_data_part_mark
        if dataset_name not in ("cityscapes", "lvis"):
            return                                    # the reference scans nothing for other names
        for entry in os.scandir(root):
            if not (entry.name.endswith(".Y.pt") and entry.is_file()):
                continue
            stem = entry.name.split(".")[0].split("_")
            if dataset_name == "cityscapes":
                caty, cid, kid, itemkey, fpos, ixhxw = stem
                fname_X = f"{caty}_{cid}_{kid}_{itemkey}_{fpos}_3x{ixhxw[2:]}.uint8.X.pt"
                idx_H, idx_W = (int(v) for v in fpos.split("x"))
                self.data_info.append({"fpath_Y": os.path.join(root, entry.name), "fpath_X": os.path.join(root, fname_X), "idx_H": idx_H,
                                       "idx_W": idx_W, "Y_cls_s": CITYSCAPES_IDX.get(int(kid[1:]), 19)})
            else:
                caty, cid, kid, aid, imgid, fpos, paddings, ixhxw = stem
                pad_left, pad_right, pad_top, pad_bottom = (int(v) for v in paddings.split("x"))
                idx_H, idx_W = (int(v) for v in fpos.split("x"))
                fpath_img = os.path.join(self.coco_path, imgid + ".jpg")
                if not os.path.exists(fpath_img):
                    for d in (coco["train2017"], coco["val2017"], coco["test2017"]):
                        fpath_img = os.path.join(d, imgid + ".jpg")
                        if os.path.exists(fpath_img):
                            break
                self.data_info.append({"fpath_Y": os.path.join(root, entry.name), "fpath_X": fpath_img, "idx_H": idx_H, "idx_W": idx_W,
                                       "Y_cls_s": int(kid[1:]), "pad_left": pad_left, "pad_right": pad_right, "pad_top": pad_top,
                                       "pad_bottom": pad_bottom})

    def __len__(self):
        return len(self.data_info)

    def __getitem__(self, idx):
        info = self.data_info[idx]
        mask = torch.load(info["fpath_Y"], weights_only=True)
        mask = mask.reshape(mask.shape[-2], mask.shape[-1]).to(torch.uint8)
        if self.dataset_name == "cityscapes":
            img = torch.load(info["fpath_X"], weights_only=True).to(torch.uint8).permute(1, 2, 0).contiguous()      # (3,H,W) -> HWC
            pads = (0, 0, 0, 0)
        else:
            from PIL import Image
            img = np.array(Image.open(info["fpath_X"]).convert("RGBA"), dtype=np.uint8)      # a writable copy of the decoded pixels
            pads = (info["pad_left"], info["pad_right"], info["pad_top"], info["pad_bottom"])
        return Sample(img, mask, pads, (info["idx_H"], info["idx_W"]), (self.HC, self.WC), info["Y_cls_s"])

    def batches(self, batch_size, indices=None):
        """Lists of decoded samples, the iterable `DevicePrefetcher` consumes (indices e.g. from train.shard_indices)."""
        idx = list(range(len(self))) if indices is None else list(indices)
        for i in range(0, len(idx), batch_size):
            yield [self[j] for j in idx[i:i + batch_size]]
