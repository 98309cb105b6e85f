"""PCIe-inclusive throughput of the training step: every step consumes a NEW batch that starts in host memory as decoded uint8
samples (RGBA image + mask, the format PreprocessDataset hands out) and reaches the device through data.DevicePrefetcher
(pinned staging -> async H2D on a copy stream -> fs_ingest_sample), overlapped with the previous step.

    python tools/host_fed_bench.py [--steps 10] [--batch 64] [--size 1024]

Prints img/s for (a) the resident synthetic batch of bench.py and (b) the host-fed loop, same model, same step.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    args = ap.parse_args()
    import fovealseg
    from fovealseg import train as T, data, ops
    fovealseg.hip.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    ops.DropoutState.seed = 1234
    B, H = args.batch, args.size
    rng = np.random.default_rng(0)
    # two distinct host batches, alternated (decode is not part of this measurement); tensors pinned once, as a loader's
    # pin_memory thread would deliver them
    host_batches = []
    for k in range(2):
        samples = []
        for i in range(B):
            img = torch.from_numpy(rng.integers(0, 256, (H, H, 4), dtype=np.uint8)).pin_memory()
            yy, xx = np.mgrid[0:H, 0:H]
            cy, cx = rng.integers(H // 4, 3 * H // 4, 2)
            mask = torch.from_numpy((((yy - cy) ** 2 + (xx - cx) ** 2) <= (0.15 * H) ** 2).astype(np.uint8)).pin_memory()
            samples.append(data.Sample(img, mask, (0, 0, 0, 0), (int(cy), int(cx)), (H, H), int(rng.integers(0, 50))))
        host_batches.append(samples)

    def feed(n):
        for i in range(n):
            yield host_batches[i % 2]

    res = {"batch": B, "size": H, "h2d_mb_per_batch": round(B * H * H * 5 / 1e6, 1)}
    # (a) resident batch
    batch = T.synthetic_batch(B, H, H, seed=1, device=dev)
    for i in range(args.warmup):
        T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=i)
    torch.cuda.synchronize()
    res["resident_img_per_s"] = round(B * args.steps / (time.perf_counter() - t0), 1)
    # (b) host-fed
    it = data.DevicePrefetcher(feed(args.warmup + args.steps), dev, channels=4)
    n = 0
    for bt in it:
        if n == args.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        T.train_step(module, optimizers, bt, cfg, epoch=1, cur_iter=n)
        n += 1
    torch.cuda.synchronize()
    res["host_fed_img_per_s"] = round(B * args.steps / (time.perf_counter() - t0), 1)
    res["ratio"] = round(res["host_fed_img_per_s"] / res["resident_img_per_s"], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
