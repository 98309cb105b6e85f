"""The C restatement (oracle/grid_sample_ref.c) against the reference's golden vectors: bit-exact."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libfs_oracle_c.so"))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("tag", ["128x128", "200x136"])
def test_c_grid_sample_bitexact(golden, clib, tag):
    g = golden("g5_gridsample_" + tag)
    x, y, grid = (np.ascontiguousarray(g[k]) for k in ("x", "y", "grid"))
    B, C, H, W = x.shape
    h, w = grid.shape[1:3]
    out = np.empty((B, C, h, w), np.float32)
    clib.fs_oracle_grid_sample(_p(x), _p(grid), _p(out), B, C, H, W, h, w)
    assert np.array_equal(out, g["x_sampled"])
    lab = np.empty((B, h, w), np.int64)
    clib.fs_oracle_label_map(_p(y), _p(grid), _p(lab), B, H, W, h, w)
    assert np.array_equal(lab, g["label"])


@pytest.mark.parametrize("H", [128, 640])
def test_c_inverse_index(golden, clib, H):
    g = golden(f"g6_inverse_{H}")
    grid = np.ascontiguousarray(g["grid"])
    n = grid.size // 2
    u = np.empty(grid.shape[:-1], np.int64)
    v = np.empty_like(u)
    clib.fs_oracle_inverse_index(_p(grid), _p(u), _p(v), ctypes.c_long(n), H, H)
    assert np.array_equal(u, g["u"]) and np.array_equal(v, g["v"])
