"""Name-keyed deterministic weight initialisation.

Every state_dict entry is filled from ``numpy.random.RandomState(crc32(name))`` over its *logical*
shape, so the reference modules (in the golden generator), the CPU oracle and the HIP modules get
bit-identical weights from the key names alone and no 522 MB state_dict is ever stored
("Goldens to generate").
"""
import zlib

import numpy as np
import torch


def _rs(name: str) -> np.random.RandomState:
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)


def name_keyed_tensor(name: str, shape, dtype=torch.float32) -> torch.Tensor:
    """Value for the state_dict entry `name` with logical `shape`."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    rs = _rs(name)
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "_running_iter":
        return torch.ones(shape, dtype=dtype)
    if leaf == "_tmp_running_mean":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "_tmp_running_var":
        return torch.ones(shape, dtype=dtype)
    z = rs.standard_normal(size=shape if len(shape) else None)
    z = np.asarray(z, dtype=np.float64).reshape(shape)
    if leaf == "running_mean":
        v = 0.05 * z
    elif leaf == "running_var":
        v = 1.0 + 0.1 * np.abs(z)
    elif leaf == "weight" and len(shape) == 1:      # BatchNorm gamma
        v = 0.5 + 0.05 * z
    elif leaf == "bias":
        v = 0.05 * z
    elif leaf == "weight" and len(shape) >= 2:      # conv / linear
        fan_in = int(np.prod(shape[1:]))
        v = z * np.sqrt(1.0 / fan_in)
    else:
        v = 0.05 * z
    return torch.from_numpy(np.ascontiguousarray(v.astype(np.float32))).to(dtype)


@torch.no_grad()
def apply_name_keyed_init(module: torch.nn.Module, prefix: str = "") -> None:
    """Overwrite every parameter and buffer of `module` in place (strided parameters are filled
    through their logical view, so the physical layout does not matter)."""
    for name, t in module.state_dict(keep_vars=True).items():
        if name.endswith("filter.weight"):        # fixed Gaussian, not a learned weight
            continue
        src = name_keyed_tensor(prefix + name, t.shape, t.dtype if t.is_floating_point() else torch.float32)
        t.data.copy_(src.to(t.dtype))
