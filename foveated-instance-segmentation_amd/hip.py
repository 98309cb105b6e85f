"""ctypes binding of libfovealseg_hip.so (C ABI: include/fovealseg.h).

There is NO fallback: if the library is missing or a launch is rejected the call raises.  torch is
used only for device memory (tensor.data_ptr()) and the current HIP stream.
"""
import ctypes
import threading
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FS_HIP_LIB") or os.path.join(_HERE, "csrc", "libfovealseg_hip.so")   # FS_HIP_LIB: kernel experiments

_P, _I, _L, _F, _U = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint32

# name -> argument type string (p pointer, i int, l long, f float, u uint32); the trailing stream
# pointer is appended automatically.
SIGNATURES = {
    "fs_ingest_sample": "ppppiiiiiiiii",
    "fs_gaze_lowres_fwd": "pppiiiii",
    "fs_compress_fwd": "ppppiii",
    "fs_compress_bwd": "ppppppiiip",
    "fs_compress_softmax_fwd": "ppppiii",
    "fs_compress_softmax_bwd": "pppppppiiip",
    "fs_area_pool_fwd": "ppiiiii",
    "fs_edge_loss_fwd": "pplfpp",
    "fs_edge_loss_bwd": "pplfppp",
    "fs_gauss_grid_fwd": "pppiiii",
    "fs_gauss_grid_bwd": "ppppiiii" + "p",
    "fs_gauss_grid_fwd_mode": "pppiiiii",
    "fs_gauss_grid_bwd_mode": "ppppiiiii" + "p",
    "fs_grid_upsample_fwd": "ppiiiii",
    "fs_grid_upsample_bwd": "ppiiiii",
    "fs_grid_sample_fwd": "pppiiiiiii",
    "fs_grid_sample_label": "ppppiiiii",
    "fs_grid_sample_bwd_grid": "ppppiiiiiii",
    "fs_grid_sample_bwd_input": "pppiiiiiii",
    "fs_inverse_index_maps": "ppplii",
    "fs_inverse_grid": "pppiiiii",
    "fs_fill_nearest": "pppiiii",
    "fs_conv2d_fwd": "ppppiiiiiiiiiiiifuplp",
    "fs_conv2d_fwd_residual": "ppppp" + "iiiiiiiiiiii" + "fufu" + "l" + "plp",
    "fs_conv2d_fwd_stats": "pppppiiiiiiiiiiiifuplp",
    "fs_conv2d_fwd_affine_act": "ppppppp" + "iiiiiiiiiiii" + "i" + "plp",
    "fs_conv2d_bwd_data": "pppiiiiiiiiiiiiplp",
    "fs_conv2d_bwd_data_bnsum": "pppiiiiiiiiiiiiplp" + "ppppp" + "pp",
    "fs_weight_amax_segments": "pppip",
    "fs_conv2d_pack": "p" + "iiiiiiiiiiii" + "i" + "plp",
    "fs_conv2d_bwd_weight": "pppiiiiiiiiiiiii" + "pl",
    "fs_linear_bwd_weight_bias": "pppp" + "lii" + "ii" + "pl",
    "fs_bn_stats": "pliffppppp",
    "fs_bn_finalize_slab": "piliffpppp",
    "fs_bn_eval_prepare": "ppifpp",
    "fs_bn_eval_affine": "ppppifpp",
    "fs_bn_act_fwd": "pppppppplii",
    "fs_bn_bwd_partial": "ppppppliip",
    "fs_add_n_bnsum": "pppppppppliip",
    "fs_bn_bwd_finalize": "pippplii" + "pppi",
    "fs_bn_bwd_apply": "ppppp" + "lii" + "fu" + "pp",
    "fs_hr_fuse_fwd": "pppipiiiii",
    "fs_relu_bwd": "pppl",
    "fs_add_n": "pppppl",
    "fs_upsample_slice_fwd": "piiiipiiii",
    "fs_upsample_slice_bwd": "piiiiipiii",
    "fs_upsample_slice_bwd_bnsum": "piiiiipiii" + "pppp",
    "fs_relu_bwd_bnsum": "ppplii" + "pppp",
    "fs_colsum": "plipip",
    "fs_maxpool_fwd": "pppiiiiiiiii",
    "fs_maxpool_bwd": "pppiiiiiiiii",
    "fs_dropout": "pplfu",
    "fs_avgpool_fwd": "piiip",
    "fs_avgpool_bwd": "piiip",
    "fs_mask_head_fwd": "ppppli",
    "fs_mask_head_bwd": "pppppppli" + "p",
    "fs_pred_assemble_fwd": "pppiii",
    "fs_pred_assemble_bwd": "pppppiii",
    "fs_seg_loss_fwd": "ppiiiffppp",
    "fs_seg_loss_bwd": "pppppiiif",
    "fs_adam_step": "pppplfffffif",
    "fs_layernorm_fwd": "pppppplif",
    "fs_layernorm_bwd": "pppppppplii" + "p",
    "fs_layernorm_bwd_add": "ppppppppplii" + "p",
    "fs_gelu_fwd": "ppl",
    "fs_gelu_bwd": "pppl",
    "fs_unfold": "pp" + "iiiiiiiiii",
    "fs_fold": "pp" + "iiiiiiiiii",
    "fs_gelu_dropout_fwd": "pplfu",
    "fs_gelu_dropout_bwd": "ppplfu",
    "fs_dwconv3_fwd": "ppppiiiii",
    "fs_dwconv3_bwd_weight": "ppppiiiii",
    "fs_dwconv3_bwd_weight_bias": "pppppiiiiii",
    "fs_residual_droppath": "pppllfu",
    "fs_droppath_dropout_bwd": "ppllfufu",
    "fs_attention_fwd": "pppppiiiiffu",
    "fs_attention_fwd_split": "ppppppp" + "l" + "iiii" + "ffu",
    "fs_attention_bwd": "ppppppppppiiiiffu",
    "fs_attention_bwd_split": "ppppppppppp" + "pl" + "iiii" + "ffu",
    "fs_attention_bwd_dq_split": "pppppppp" + "pl" + "iiii" + "ffu",
    "fs_attention_bwd_dkv_split": "pppppppppp" + "iiii" + "ffu",
}
_CT = {"p": _P, "i": _I, "l": _L, "f": _F, "u": _U}

_lib = None
# declared in the header, host-side only (no stream argument)
HOST_ONLY = ("fs_set_conv_precision", "fs_get_conv_precision", "fs_conv2d_workspace_bytes", "fs_conv2d_stats_slabs", "fs_conv2d_kernel_choice",
             "fs_bn_bwd_slabs", "fs_dwconv3_wgrad_lanes", "fs_conv2d_bwd_data_bnsum_slabs",
             "fs_conv2d_fwd_affine_act_ok", "fs_conv2d_fwd_residual_ok", "fs_linear_bwd_weight_bias_ok", "fs_attention_split_ws_bytes", "fs_attention_bwd_split_ws_bytes", "fs_attention_mask_words", "fs_attention_bwd_split_parts_offset",
             "fs_stream_wait", "fs_set_deterministic", "fs_get_deterministic", "fs_conv2d_bwd_weight_ws_bytes", "fs_linear_bwd_weight_bias_ws_bytes",
             "fs_colsum_scratch_floats", "fs_bn_stats_scratch_doubles", "fs_mask_head_bwd_scratch_floats", "fs_layernorm_bwd_scratch_floats",
             "fs_conv2d_pack_persistent", "fs_conv2d_ws_mode",
             "fs_edge_loss_stats_floats", "fs_compress_softmax_bwd_scratch_floats", "fs_gauss_grid_bwd_scratch_floats")


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the library.  Raises if it is missing or was built from other sources than the ones beside it; never builds, never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU/eager fallback for the fovealseg path.")
    if not os.environ.get("FS_HIP_LIB"):
        # a stale binary cannot load: the library carries the content hash of the sources it was built from (build.py)
        from . import build as _build
        have, want = _build.built_hash(LIB_PATH), _build.source_hash()
        if have != want:
            raise HipLibraryError(
                f"{LIB_PATH} was built from other sources than the csrc/ beside it (stamp {str(have)[:12]}, sources {want[:12]}): "
                "run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = _I
        fn.argtypes = [_CT[c] for c in sig] + [_P]
    lib.fs_set_conv_precision.restype = _I
    lib.fs_set_conv_precision.argtypes = [_I]
    lib.fs_get_conv_precision.restype = _I
    lib.fs_get_conv_precision.argtypes = []
    lib.fs_colsum_scratch_floats.restype = _L
    lib.fs_colsum_scratch_floats.argtypes = [_L, _I]
    lib.fs_bn_stats_scratch_doubles.restype = _L
    lib.fs_bn_stats_scratch_doubles.argtypes = [_L, _I]
    lib.fs_mask_head_bwd_scratch_floats.restype = _L
    lib.fs_mask_head_bwd_scratch_floats.argtypes = [_L, _I]
    lib.fs_layernorm_bwd_scratch_floats.restype = _L
    lib.fs_layernorm_bwd_scratch_floats.argtypes = [_L, _I]
    lib.fs_edge_loss_stats_floats.restype = _L
    lib.fs_edge_loss_stats_floats.argtypes = [_L]
    lib.fs_compress_softmax_bwd_scratch_floats.restype = _L
    lib.fs_compress_softmax_bwd_scratch_floats.argtypes = [_I, _I]
    lib.fs_gauss_grid_bwd_scratch_floats.restype = _L
    lib.fs_gauss_grid_bwd_scratch_floats.argtypes = [_I, _I, _I]
    lib.fs_stream_wait.restype = _I
    lib.fs_stream_wait.argtypes = [_P, _P]
    lib.fs_set_deterministic.restype = _I
    lib.fs_set_deterministic.argtypes = [_I]
    lib.fs_get_deterministic.restype = _I
    lib.fs_get_deterministic.argtypes = []
    lib.fs_conv2d_bwd_weight_ws_bytes.restype = _L
    lib.fs_conv2d_bwd_weight_ws_bytes.argtypes = [_I] * 7
    lib.fs_linear_bwd_weight_bias_ws_bytes.restype = _L
    lib.fs_linear_bwd_weight_bias_ws_bytes.argtypes = [_I] * 2
    lib.fs_conv2d_workspace_bytes.restype = _L
    lib.fs_conv2d_workspace_bytes.argtypes = [_I] * 12
    lib.fs_conv2d_stats_slabs.restype = _I
    lib.fs_conv2d_stats_slabs.argtypes = [_I] * 12 + [_L]
    lib.fs_conv2d_kernel_choice.restype = _I
    lib.fs_conv2d_kernel_choice.argtypes = [_I] * 13 + [_L]
    lib.fs_conv2d_pack_persistent.restype = _I
    lib.fs_conv2d_pack_persistent.argtypes = [_I] * 13 + [_L]
    lib.fs_conv2d_ws_mode.restype = _I
    lib.fs_conv2d_ws_mode.argtypes = [_I]
    lib.fs_bn_bwd_slabs.restype = _I
    lib.fs_bn_bwd_slabs.argtypes = [_L, _I]
    lib.fs_dwconv3_wgrad_lanes.restype = _I
    lib.fs_dwconv3_wgrad_lanes.argtypes = [_I] * 4
    lib.fs_conv2d_bwd_data_bnsum_slabs.restype = _I
    lib.fs_conv2d_bwd_data_bnsum_slabs.argtypes = [_I] * 12 + [_L]
    lib.fs_conv2d_fwd_affine_act_ok.restype = _I
    lib.fs_conv2d_fwd_affine_act_ok.argtypes = [_I] * 12 + [_L]
    lib.fs_conv2d_fwd_residual_ok.restype = _I
    lib.fs_conv2d_fwd_residual_ok.argtypes = [_I] * 12 + [_L, _L]
    lib.fs_linear_bwd_weight_bias_ok.restype = _I
    lib.fs_linear_bwd_weight_bias_ok.argtypes = [_L, _I, _I]
    lib.fs_attention_split_ws_bytes.restype = _L
    lib.fs_attention_split_ws_bytes.argtypes = [_I] * 3
    lib.fs_attention_bwd_split_ws_bytes.restype = _L
    lib.fs_attention_bwd_split_ws_bytes.argtypes = [_I] * 3
    lib.fs_attention_bwd_split_parts_offset.restype = _L
    lib.fs_attention_bwd_split_parts_offset.argtypes = [_I] * 3
    lib.fs_attention_mask_words.restype = _L
    lib.fs_attention_mask_words.argtypes = [_I] * 4
    _lib = lib
    global _default_mode
    _default_mode = ("f32", "bf16x3", "f16x2")[lib.fs_get_conv_precision()]
    return lib


_default_mode = None


def default_conv_precision() -> str:
    """The mode the library came up in: 'bf16x3' (24-bit operands, the reference's fp32 operand width) unless FS_CONV_PRECISION
    named another one before the library was loaded."""
    load()
    return _default_mode


def set_conv_precision(mode: str) -> None:
    """'f32' (fp32 MFMA), 'bf16x3' (three bf16 terms per operand, six bf16 MFMAs per product) or 'f16x2' (two scaled
    fp16 terms, three fp16 MFMAs per product).  All three accumulate in fp32 and stay within the rounding error of an fp32
    accumulation chain; layers whose channel counts are no multiple of 4 run on the fp32-MFMA kernels in every mode."""
    code = {"f32": 0, "bf16x3": 1, "f16x2": 2}[mode]
    if load().fs_set_conv_precision(code) != 0:
        raise HipLibraryError("fs_set_conv_precision rejected the mode")
    global _mode_name
    _mode_name = mode
    _ws_cache.clear()


def set_deterministic(on: bool) -> None:
    """Bit-reproducible training steps (the reference's cudnn.deterministic = True): bwd-weight sums its split-K partial tiles in index
    order instead of by fp32 atomics (include/fovealseg.h fs_set_deterministic)."""
    if load().fs_set_deterministic(1 if on else 0) != 0:
        raise HipLibraryError("fs_set_deterministic failed")
    _ws_cache.clear()


def get_deterministic() -> bool:
    return bool(load().fs_get_deterministic())


_ws_cache = {}


def query(name, *args):
    """A host-only size / predicate query of the library (fs_*_bytes, fs_*_floats ...), cached per argument tuple: one dictionary lookup
    instead of a ctypes call on every launch.  The cache is dropped whenever the precision or deterministic mode changes."""
    key = (name,) + args
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(getattr(load(), name)(*args))
    return v


def wgrad_workspace(device, Cin, Cout, R, S, stride, pad, dil):
    """(scratch tensor | None, bytes) for fs_conv2d_bwd_weight: per-split partial tiles (deterministic mode; strided 3x3 layers in every mode)."""
    n = query("fs_conv2d_bwd_weight_ws_bytes", Cin, Cout, R, S, stride, pad, dil)
    if n == 0:
        return None, 0
    return torch.empty(n, device=device, dtype=torch.uint8), n


def conv_workspace_bytes(H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed):
    """Scratch bytes the conv entry points can use for this shape under the current precision mode (cached)."""
    key = (H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_conv2d_workspace_bytes(*key))
    return v


def conv_stats_slabs(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes):
    key = ("slabs", B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_conv2d_stats_slabs(*key[1:]))
    return v


def attention_split_ws_bytes(B, Nk, heads, backward=False):
    """Scratch bytes of the split-precision attention entry points (packed K / V planes)."""
    return query("fs_attention_bwd_split_ws_bytes" if backward else "fs_attention_split_ws_bytes", B, Nk, heads)


def linear_bwd_weight_bias_ok(rows, Cin, Cout):
    """True where fs_linear_bwd_weight_bias (dW and dbias of a linear layer in one launch) exists under the current precision mode."""
    key = ("lwb", rows, Cin, Cout)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_linear_bwd_weight_bias_ok(rows, Cin, Cout)) == 1
    return v


def fwd_affine_act_ok(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes):
    """True where fs_conv2d_fwd_affine_act serves this shape under the current precision mode (cached)."""
    key = ("faa", B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_conv2d_fwd_affine_act_ok(*key[1:]))
    return v == 1


def fwd_residual_ok(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, rows_per_sample, ws_bytes):
    """True where fs_conv2d_fwd_residual serves this shape under the current precision mode (cached)."""
    key = ("fres", B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, rows_per_sample, ws_bytes)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_conv2d_fwd_residual_ok(*key[1:]))
    return v == 1


def bwd_data_bnsum_slabs(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes):
    """Slab rows of fs_conv2d_bwd_data_bnsum for this problem under the current precision mode, 0 = not available (cached)."""
    key = ("bdsum", B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_conv2d_bwd_data_bnsum_slabs(*key[1:]))
    return v


def bn_bwd_slabs(M, C):
    """Rows of the [.][C][2] partial-sum slab the BatchNorm-backward producers write for an (M, C) activation (cached)."""
    key = ("bnslab", M, C)
    v = _ws_cache.get(key)
    if v is None:
        v = _ws_cache[key] = int(load().fs_bn_bwd_slabs(M, C))
    return v


_mode_name = None      # the library's precision mode as a string, kept in step by set_conv_precision (one ctypes call less per conv launch)


def get_conv_precision() -> str:
    global _mode_name
    if _mode_name is None:
        _mode_name = ("f32", "bf16x3", "f16x2")[load().fs_get_conv_precision()]
    return _mode_name


# torch's current HIP stream as a raw handle.  torch.cuda.current_stream() builds a Python Stream object through several layers of device
# index helpers (~10 us, twice the cost of the ctypes launch itself: configs[4] is host-bound at ~1 700 launches per 30 ms step); the
# private accessor returns the same handle in ~0.3 us and follows stream guards and the autograd engine's per-node streams alike.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_fn_cache = {}


# A raw stream handle: the calling THREAD's launches go there instead of torch's current stream (ops: side-stream weight gradients, the
# weight-pack prefetch).  Per thread (round 5): a ctypes call releases the GIL, so a second launching thread -- a prefetch thread, a
# serving worker -- must never see another thread's override.
_tls = threading.local()


def stream_override():
    return getattr(_tls, "stream", None)


def set_stream_override(handle):
    """Route this thread's launches to `handle` (None: back to torch's current stream); returns the previous value."""
    old = getattr(_tls, "stream", None)
    _tls.stream = handle
    return old


def stream_wait(waiter, signaller):
    """`waiter` (raw handle) waits for everything enqueued so far on `signaller` (fs_stream_wait: one host call)."""
    lib = _lib if _lib is not None else load()
    err = lib.fs_stream_wait(waiter, signaller)
    if err != 0:
        raise HipLibraryError(f"fs_stream_wait: hipError {err}")


def call_packed(name, *args):
    """`call` for a conv entry point whose scratch already holds this layer's weight pack (fs_conv2d_ws_mode, ops.PackCache)."""
    lib = _lib if _lib is not None else load()
    lib.fs_conv2d_ws_mode(1)
    try:
        call(name, *args)
    finally:
        lib.fs_conv2d_ws_mode(0)


def call(name, *args):
    """Launch `name` on torch's current HIP stream (or this thread's stream override); raises on any non-zero status."""
    fn = _fn_cache.get(name)
    if fn is None:
        lib = _lib if _lib is not None else load()
        fn = _fn_cache[name] = getattr(lib, name)
    so = getattr(_tls, "stream", None)
    err = fn(*args, so if so is not None else _stream())
    if err != 0:
        what = "argument rejected at the C-ABI boundary" if err == 1001 else f"hipError {err}"
        raise HipLibraryError(f"{name}: {what}")


def ptr(t):
    """Device pointer of a tensor the kernels may touch: fp32/fp64/int64, CUDA(HIP), contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("fovealseg kernels need device tensors (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise HipLibraryError(f"non-contiguous tensor {tuple(t.shape)} strides {t.stride()} passed to a HIP kernel")
    return t.data_ptr()
