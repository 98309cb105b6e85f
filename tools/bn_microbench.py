"""Per-launch time and HBM rate of the BN forward / backward C-ABI calls on the HRNet tensor shapes.

    python tools/bn_microbench.py [reps]

Bytes are algorithmic + the re-reads of the two-pass backward (what the kernels actually move), so GB/s is the HBM rate.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fovealseg
from fovealseg import hip

SHAPES = [(409600, 64), (102400, 128), (25600, 256), (6400, 512), (409600, 256), (409600, 192), (409600, 240)]


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
    hip.load()
    dev = "cuda"
    for M, C in SHAPES:
        y = torch.randn(M, C, device=dev)
        res = torch.randn(M, C, device=dev)
        dz = torch.randn(M, C, device=dev)
        z = torch.empty_like(y); dy = torch.empty_like(y); dres = torch.empty_like(y)
        mask = torch.empty(M * C // 4, device=dev, dtype=torch.uint8)
        mean = y.mean(0).contiguous(); invstd = (1.0 / y.std(0)).contiguous()
        gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev)
        dgamma = torch.empty(C, device=dev); dbeta = torch.empty(C, device=dev)
        nslab = hip.bn_bwd_slabs(M, C)
        slab = torch.empty(nslab * C * 2, device=dev); coef = torch.empty(4 * C, device=dev)
        out = []
        for has_res in (False, True):
            r = res if has_res else None
            dr = dres if has_res else None
            tf = timed(lambda: hip.call("fs_bn_act_fwd", hip.ptr(y), hip.ptr(mean), hip.ptr(invstd), hip.ptr(gamma), hip.ptr(beta), hip.ptr(r),
                                        hip.ptr(z), hip.ptr(mask), M, C, 1), reps)
            def bwd():
                hip.call("fs_bn_bwd_partial", hip.ptr(dz), None, hip.ptr(mask), hip.ptr(y), hip.ptr(mean), hip.ptr(invstd), M, C, 1, hip.ptr(slab))
                hip.call("fs_bn_bwd_finalize", hip.ptr(slab), nslab, hip.ptr(gamma), hip.ptr(mean), hip.ptr(invstd), M, C, 1, hip.ptr(coef),
                         hip.ptr(dgamma), hip.ptr(dbeta), 0)
                hip.call("fs_bn_bwd_apply", hip.ptr(dz), None, hip.ptr(mask), hip.ptr(y), hip.ptr(coef), M, C, 1, 0.3, 77, hip.ptr(dy), hip.ptr(dr))
            tb = timed(bwd, reps)
            bf = 4.0 * M * C * (2 + has_res) + M * C / 4
            bb = 4.0 * M * C * (5 + has_res) + 2 * M * C / 4
            out.append(f"res={int(has_res)} fwd {tf:6.1f} us {bf / tf / 1e3:6.0f} GB/s | bwd {tb:6.1f} us {bb / tb / 1e3:6.0f} GB/s")
        print(f"M={M} C={C}: " + " || ".join(out), flush=True)


if __name__ == "__main__":
    main()
