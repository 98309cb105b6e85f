"""Measurement of the rows next to the hot path on one MI355X, with the CPU oracle timed beside each.

    python tools/next_rows_bench.py [--batch 64] [--out profiles/r01/next_rows_bench.json]

 f-1  input pipeline : fs_ingest_sample over a batch of decoded LVIS-shaped samples (640x640x4 uint8 RGBA + mask, padded to 1024^2)
                       -- device-only (uint8 already resident) and through DevicePrefetcher (pinned H2D + convert, PCIe-inclusive)
 f-2  metrics        : DeviceMeter.update per step (no host sync) against 3 x .item()
 f-3  inverse warp   : fs_inverse_grid, fs_grid_sample_fwd through the inverse grid, fs_fill_nearest at 80^2 -> 1024^2, 51 classes
All three kernels families are HBM-bound byte movers; GB/s is algorithmic bytes (each operand read once, each result written
once) over the HIP-event time, against the 8 TB/s HBM peak.  The oracle is used here as the timed CPU baseline only.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

PEAK_HBM_GBS = 8000.0


def gpu_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def hbm(bytes_, secs):
    g = bytes_ / secs / 1e9
    return {"seconds": round(secs, 6), "algorithmic_mb": round(bytes_ / 1e6, 2), "gb_per_s": round(g, 1), "frac_of_hbm_peak": round(g / PEAK_HBM_GBS, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import fovealseg
    from fovealseg import hip, ops, data, train as T
    import fovealseg_oracle as O
    hip.load()
    dev = torch.device("cuda", 0)
    B = args.batch
    rng = np.random.default_rng(3)
    res = {"device": torch.cuda.get_device_name(0), "batch": B}

    # ---- f-1 ------------------------------------------------------------------------------------
    H = W = 640
    pads = (192, 192, 192, 192)                    # -> 1024 x 1024, the bench's frame
    samples = [data.Sample(torch.from_numpy(rng.integers(0, 256, (H, W, 4), dtype=np.uint8)), torch.from_numpy(rng.integers(0, 2, (H, W), dtype=np.uint8)),
                           pads, (300 + i, 310 + i), (1024, 1024), i % 50) for i in range(B)]
    staged = [(s.img.to(dev), s.mask.to(dev)) for s in samples]
    HP, WP = samples[0].padded_hw
    by = B * (H * W * 5 + HP * WP * 5 * 4)         # uint8 RGBA+mask read, fp32 (4+1, HP, WP) written
    t = gpu_time(lambda: data.ingest_batch(samples, dev, channels=4, staged=staged), args.reps)
    res["f1_ingest_device"] = dict(hbm(by, t), img_per_s=round(B / t, 1), what=f"{B} x fs_ingest_sample + F/cls upload, uint8 {H}x{W}x4 (RGBA) + mask -> fp32 (4+1,{HP},{WP})")
    nb = 4
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in data.DevicePrefetcher(iter([samples] * nb), dev, channels=4):
        n += batch[0].shape[0]
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    res["f1_prefetcher_pcie_inclusive"] = {"img_per_s": round(n / t, 1), "h2d_mb_per_batch": round(B * H * W * 5 / 1e6, 1),
                                           "fp32_h2d_mb_per_batch_reference": round(B * HP * WP * 5 * 4 / 1e6, 1),
                                           "what": f"{nb} batches: pin + async H2D of the uint8 samples + fs_ingest_sample, wall clock"}
    t0 = time.perf_counter()
    for s in samples[:8]:
        O.ingest_sample_ref(s.img.numpy(), s.mask.numpy(), s.pads, s.focus, s.frame, s.cls)
    t = (time.perf_counter() - t0) / 8
    res["f1_cpu_oracle"] = {"img_per_s": round(1.0 / t, 1), "what": "ingest_sample_ref (ToTensor + F.pad), 8 samples, 1 process"}

    # ---- f-2 ------------------------------------------------------------------------------------
    vals = [torch.rand((), device=dev) for _ in range(3)]
    meter = T.DeviceMeter(["loss", "acc", "edge"], dev)
    for _ in range(20):                                # first calls load the torch kernels
        meter.update(vals)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        meter.update(vals)
    th = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        [v.item() for v in vals]
    ti = (time.perf_counter() - t0) / 200
    res["f2_meter"] = {"device_meter_host_us_per_step": round(th * 1e6, 1), "three_item_calls_us_per_step_idle_gpu": round(ti * 1e6, 1),
                       "what": "host cost per step; .item() additionally drains the queue (a full step, ~170 ms, when the GPU is busy)"}

    # ---- f-3 ------------------------------------------------------------------------------------
    Bu, C, h, w, Hs, Ws = 16, 51, 80, 80, 1024, 1024
    g = torch.Generator().manual_seed(5)
    base = torch.stack(torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")[::-1], dim=-1)
    grid = (base[None].sign() * base[None].abs() ** 1.4 + 0.004 * torch.randn(Bu, h, w, 2, generator=g)).clamp(-1, 1).contiguous()
    pred = torch.randn(Bu, C, h, w, generator=g)
    gd, pd = grid.to(dev), pred.to(dev)
    owner = torch.empty(Bu, Hs, Ws, device=dev, dtype=torch.int32)
    inv = torch.empty(Bu, Hs, Ws, 2, device=dev, dtype=torch.float32)
    out = torch.empty(Bu, C, Hs, Ws, device=dev, dtype=torch.float32)
    scratch = torch.empty(2 * Bu * Hs * Ws, device=dev, dtype=torch.int32)
    px = Bu * Hs * Ws
    t = gpu_time(lambda: hip.call("fs_inverse_grid", hip.ptr(gd), hip.ptr(owner), hip.ptr(inv), Bu, h, w, Hs, Ws), args.reps)
    res["f3_inverse_grid"] = dict(hbm(px * 12 + gd.numel() * 4, t), what=f"owner int32 + grid_inv 2 x fp32 written for {Bu} x {Hs}x{Ws}")
    t = gpu_time(lambda: hip.call("fs_grid_sample_fwd", hip.ptr(pd), hip.ptr(inv), hip.ptr(out), Bu, C, h, w, Hs, Ws, 0), args.reps)
    res["f3_grid_sample_through_inverse"] = dict(hbm(px * 8 + px * C * 4 + pd.numel() * 4, t), what=f"{C} classes, {h}x{w} -> {Hs}x{Ws}")
    t = gpu_time(lambda: hip.call("fs_fill_nearest", hip.ptr(out), hip.ptr(owner), hip.ptr(scratch), Bu, C, Hs, Ws), args.reps)
    holes = float((owner < 0).float().mean())
    res["f3_fill_nearest"] = dict(hbm(px * 4 + 2 * holes * px * C * 4, t), hole_fraction=round(holes, 4),
                                  what="row pass + column pass over the owner map, then one read + one write per hole pixel and class")
    t = gpu_time(lambda: ops.unwarp_nearest(pd, gd, Hs, Ws), max(2, args.reps // 3))
    res["f3_unwarp_total"] = {"seconds": round(t, 5), "img_per_s": round(Bu / t, 1), "what": "ops.unwarp_nearest incl. allocations"}
    # CPU route of the reference for one image: scatter inverse grid, F.grid_sample, scipy nearest-neighbour fill (interp2d.py);
    # the oracle's own brute-force fill is for small test sizes only
    import torch.nn.functional as F
    from scipy.interpolate import NearestNDInterpolator
    t0 = time.perf_counter()
    inv_ref = O.inverse_grid_ref(grid[:1], Hs, Ws)
    hole = torch.isnan(inv_ref[..., 0])[0].numpy()
    o = F.grid_sample(pred[:1], torch.nan_to_num(inv_ref, nan=0.0), align_corners=False)[0].numpy()
    ys, xs = np.nonzero(~hole)
    hy, hx = np.nonzero(hole)
    f = NearestNDInterpolator(np.stack([ys, xs], 1), o[:, ys, xs].T)
    o[:, hy, hx] = f(np.stack([hy, hx], 1)).T
    t = time.perf_counter() - t0
    res["f3_cpu_oracle"] = {"img_per_s": round(1.0 / t, 3), "what": "1 image on this host: inverse_grid_ref + F.grid_sample + scipy NearestNDInterpolator (the reference's fill route)"}

    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
