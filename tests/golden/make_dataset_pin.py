"""Pin of the input-pipeline row taken from the reference's own `PreprocessDataset`
(DynamicFocus/e_preprocess_scripts/dataset.py:44-142): a small synthetic LVIS-style tree is written to a temp directory, the
REFERENCE class is instantiated on it (file-name parsing, COCO image lookup with fallback, padding / focus / class-id handling)
and its `data_info` plus the samples `__getitem__` returns are stored as data in `g16_dataset.json` / `g16_dataset.npz`.

torchvision is absent, so `T.ToTensor()` is a stand-in (PIL image -> uint8 HWC -> CHW float / 255, torchvision's documented
behaviour for uint8 images): everything AROUND it -- RGBA conversion, F.pad order (left, right, top, bottom), F_2 = idx / 640,
int64 class id, dtypes -- is the reference's code.  Run once in the build container:  python tests/golden/make_dataset_pin.py
"""
import importlib
import importlib.machinery
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# the synthetic tree: (caty, cid, kid, aid, imgid, fpos, paddings, IxHxW) of every mask file, image sizes follow from H, W
TREE = [
    ("lvis", "c12", "k7", "a101", "000000000139", "200x310", "107x107x0x0", "1x640x426", "train2017"),
    ("lvis", "c3", "k49", "a102", "000000000285", "5x630", "0x0x53x53", "1x534x640", "train2017"),
    ("lvis", "c3", "k0", "a103", "000000000632", "320x320", "0x0x0x0", "1x640x640", "val2017"),      # image only in val2017: fallback lookup
    ("lvis", "c40", "k21", "a7", "000000000724", "639x0", "80x80x64x64", "1x512x480", "test2017"),   # ... only in test2017
    ("lvis", "c1", "k33", "a9999", "000000000776", "17x400", "1x0x0x1", "1x639x639", "train2017"),
]


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _ToTensor:
    """torchvision.transforms.ToTensor for PIL uint8 images: HWC uint8 -> CHW float32 in [0, 1] (transforms/functional.py:to_tensor)."""

    def __call__(self, pic):
        a = np.asarray(pic, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def write_tree(root):
    """Writes the mask files and the images; returns {relative path: sha-less description}.  Images are PNG bytes in *.jpg names
    (PIL opens by content), so decoding is lossless and identical everywhere."""
    cook = os.path.join(root, "data_c_cook", "lvis", "train", "sp60000")
    os.makedirs(cook)
    raw = os.path.join(root, "data_a_raw", "coco2017")
    for d in ("train2017", "val2017", "test2017"):
        os.makedirs(os.path.join(raw, d))
    for i, (caty, cid, kid, aid, imgid, fpos, pads, ixhxw, where) in enumerate(TREE):
        _, H, W = (int(v) for v in ixhxw.split("x"))
        g = np.random.default_rng(100 + i)
        img = g.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        Image.fromarray(img, "RGB").save(os.path.join(raw, where, imgid + ".jpg"), format="PNG")
        yy, xx = np.mgrid[0:H, 0:W]
        mask = (((yy - H * 0.4) ** 2 + (xx - W * 0.6) ** 2) <= (0.2 * min(H, W)) ** 2).astype(np.uint8)
        torch.save(torch.from_numpy(mask)[None], os.path.join(cook, f"{caty}_{cid}_{kid}_{aid}_{imgid}_{fpos}_{pads}_{ixhxw}.uint8.Y.pt"))
    # distractors the reference ignores
    open(os.path.join(cook, "README.txt"), "w").write("not a sample")
    os.makedirs(os.path.join(cook, "sub.Y.pt"))
    return os.path.join(root, "data_c_cook"), raw


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    _stub("torchvision")
    tr = _stub("torchvision.transforms", ToTensor=_ToTensor)
    sys.modules["torchvision"].transforms = tr
    os.environ.setdefault("HOSTNAME", "buildbox")
    ds = importlib.import_module("DynamicFocus.e_preprocess_scripts.dataset")
    with tempfile.TemporaryDirectory() as root:
        data_path, raw = write_tree(root)
        ds.dpath_data_raw_coco_train = os.path.join(raw, "train2017")
        ds.dpath_data_raw_coco_valid = os.path.join(raw, "val2017")
        ds.dpath_data_raw_coco_test = os.path.join(raw, "test2017")
        d = ds.PreprocessDataset(data_path=data_path, marker="sp60000", dataset_partition="train", dataset_name="lvis")
        infos = sorted(d.data_info, key=lambda r: r["fpath_Y"])
        table, arrays = [], {}
        for r in infos:
            idx = d.data_info.index(r)
            X, F2, Y, cls = d[idx]
            rec = {k: (os.path.relpath(v, root) if k.startswith("fpath") else v) for k, v in r.items()}
            rec.update(X_shape=list(X.shape), X_dtype=str(X.dtype), Y_shape=list(Y.shape), Y_dtype=str(Y.dtype), F2_dtype=str(F2.dtype),
                       cls_dtype=str(cls.dtype), cls=int(cls[0]), F2=[float(F2[0]), float(F2[1])],
                       X_sum=float(X.double().sum()), Y_sum=float(Y.double().sum()))
            key = os.path.basename(r["fpath_Y"]).split(".")[0]
            arrays[key + ":Xcrop"] = X[:, ::37, ::41].numpy()           # strided sample of the padded image incl. the alpha plane
            arrays[key + ":Ycrop"] = Y[:, ::37, ::41].numpy()
            arrays[key + ":F2"] = F2.numpy()
            table.append(rec)
        out = {"HC": d.HC, "WC": d.WC, "len": len(d), "tree": [list(t) for t in TREE], "data_info": table}
    with open(os.path.join(HERE, "g16_dataset.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "g16_dataset.npz"), **arrays)
    print("wrote g16_dataset.json / .npz:", len(table), "samples;", {k: v.shape for k, v in list(arrays.items())[:3]})


if __name__ == "__main__":
    main()
