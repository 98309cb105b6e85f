#!/usr/bin/env python3
"""Times the split-precision attention entry points one by one on the configs[3] stage shapes (B = 16, tokens at 160x160, Nk = 400).
Usage: python tools/attn_microbench.py [reps] [scale]   (a negative scale is a timing-only experiment switch of development builds)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import hip as H

SHAPES = [(16, 1, 25600, 400), (16, 2, 6400, 400), (16, 5, 1600, 400), (16, 8, 400, 400)]      # B, heads, N, Nk


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.125
    H.set_conv_precision("bf16x3")
    dev = "cuda"
    only = int(os.environ.get("MB_SHAPE", "-1"))
    for si, (B, heads, N, Nk) in enumerate(SHAPES):
        if only >= 0 and si != only:
            continue
        C = heads * 64
        q, k, v, go = (torch.randn(B, n, C, device=dev) for n in (N, Nk, Nk, N))
        o, lse = torch.empty_like(q), torch.empty(B * heads * N, device=dev)
        mask = torch.zeros(int(H.load().fs_attention_mask_words(B, N, Nk, heads)), device=dev, dtype=torch.int32)
        nb, nbb = H.attention_split_ws_bytes(B, Nk, heads), H.attention_split_ws_bytes(B, Nk, heads, backward=True)
        ws, wsb = torch.empty(nb, device=dev, dtype=torch.uint8), torch.empty(nbb, device=dev, dtype=torch.uint8)
        dq, dk, dv, D = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.zeros(B * heads * N, device=dev)
        p, key = 0.2, 12345
        flops = 4.0 * B * heads * N * Nk * 64
        t_f = timed(lambda: H.call("fs_attention_fwd_split", H.ptr(q), H.ptr(k), H.ptr(v), H.ptr(o), H.ptr(lse), H.ptr(mask), H.ptr(ws), nb,
                                   B, N, Nk, heads, scale, p, key), reps)
        t_q = timed(lambda: H.call("fs_attention_bwd_dq_split", H.ptr(q), H.ptr(k), H.ptr(v), H.ptr(go), H.ptr(lse), H.ptr(D), H.ptr(mask),
                                   H.ptr(dq), H.ptr(wsb), nbb, B, N, Nk, heads, scale, p, key), reps)
        parts = torch.empty(2 * 8 * B * Nk * C, device=dev)
        t_kv = timed(lambda: H.call("fs_attention_bwd_dkv_split", H.ptr(q), H.ptr(k), H.ptr(v), H.ptr(go), H.ptr(lse), H.ptr(D), H.ptr(mask),
                                    H.ptr(dk), H.ptr(dv), H.ptr(parts), B, N, Nk, heads, scale, p, key), reps)
        print(f"B{B} h{heads} N{N} Nk{Nk}: fwd {t_f:7.1f} us {flops / t_f / 1e6:6.1f} TF | dq {t_q:7.1f} us {1.5 * flops / t_q / 1e6:6.1f} TF | "
              f"dkv {t_kv:7.1f} us {2.5 * flops / t_kv / 1e6:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
