"""How long does the host take to ENQUEUE one train step, against how long the GPU takes to run it?

If enqueue time ~ step time the step is launch-bound (kernel speed-ups cannot show); if enqueue << step the GPU is the bound.
Prints per-step enqueue / total milliseconds, with the branch streams on and off.

    python tools/host_enqueue_probe.py [--steps 6] [--batch 64]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--profile", action="store_true", help="cProfile one enqueue and print the top host functions")
    ap.add_argument("--config", default="headline", help="headline | config3 (SegFormer 160x160) | config4 (DeepLab, 2048^2 input)")
    args = ap.parse_args()
    import fovealseg
    from fovealseg import train as T, ops, modules as Mods
    fovealseg.hip.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = fovealseg.lvis50_cfg()
    if args.config == "config3":
        cfg.MODEL.arch_encoder, cfg.MODEL.fc_dim = "segformer", 1024
        cfg.TRAIN.task_input_size = (160, 160)
    elif args.config == "config4":
        cfg.MODEL.arch_encoder = "deeplab"
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    batch = T.synthetic_batch(args.batch, args.size, args.size, seed=1, device=dev)
    ops.DropoutState.seed = 1234
    for par in (True, False):
        Mods.PARALLEL_BRANCHES = par
        for i in range(2):
            T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=i)
        torch.cuda.synchronize()
        for i in range(args.steps):
            t0 = time.perf_counter()
            T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=2 + i)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"branches_parallel={par} step {i}: enqueue {1e3 * (t1 - t0):7.1f} ms   total {1e3 * (t2 - t0):7.1f} ms", flush=True)
        # back-to-back (what bench.py times)
        t0 = time.perf_counter()
        for i in range(args.steps):
            T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=10 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"branches_parallel={par} back-to-back: enqueue {1e3 * (t1 - t0) / args.steps:7.1f} ms/step   total {1e3 * (t2 - t0) / args.steps:7.1f} ms/step", flush=True)
    if args.profile:
        import cProfile, pstats
        Mods.PARALLEL_BRANCHES = True
        pr = cProfile.Profile()
        pr.enable()
        T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=30)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
