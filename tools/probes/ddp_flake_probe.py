"""One red run of tests/test_ddp_gloo.py::test_reference_ddp_wrapper_line_on_gpu in eight (the saliency / compress arenas of the DDP-reducer
path 4 % off the explicit all-reduce; green again on the next two fresh boxes): this probe repeats the comparison N times inside one
2-rank spawn and prints, per trial, the arena errors, both losses and the worst parameters, to catch the discrepancy with its cause.

    python tools/probes/ddp_flake_probe.py [trials]
"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, trials):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import fovealseg
    from fovealseg import ops, train
    from torch.nn.parallel import DistributedDataParallel as DDP
    train.ddp_setup(backend="gloo")
    dev = torch.device("cuda", 0)
    cfg = fovealseg.lvis50_cfg()
    torch.manual_seed(5 + rank)
    module, nets = train.build_module(cfg, device=dev, init="random" if rank else "name_keyed")
    module.train()
    optimizers = train.create_optimizers(nets, cfg)
    train.broadcast_parameters(optimizers, module)
    batch = train.synthetic_batch(2, 256, 256, seed=11 + rank, device=dev)
    X, Fp, Y, cls = batch
    ddp = None

    def fwd_bwd(m):
        for opt in optimizers:
            opt.zero_grad()
        ops.DropoutState.seed, ops.DropoutState.step = 77 + rank, 1
        feed = {"img_data": X[:, :3], "seg_label": Y, "focus_point": Fp, "cls_label": cls}
        loss = m(feed, epoch=1, cur_iter=0)[0]
        loss.mean().backward()
        return float(loss)

    for t in range(trials):
        la = fwd_bwd(module)
        local_a = [o.flat.grad.clone() for o in optimizers]          # this rank's own gradient, explicit path
        train.allreduce_gradients(optimizers)
        want = [o.flat.grad * o.grad_scale for o in optimizers]
        if ddp is None:
            ddp = DDP(module, device_ids=[0], find_unused_parameters=True)
        lb = fwd_bwd(ddp)
        torch.cuda.synchronize()
        errs = [float((o.flat.grad - w).norm() / w.norm().clamp_min(1e-30)) for o, w in zip(optimizers, want)]
        # a third, collective-free sample of the local gradient: is the bare path reproducible on this rank?
        lc = fwd_bwd(module)
        torch.cuda.synchronize()
        ops.join_wgrad_streams()
        torch.cuda.synchronize()
        rep = [float((o.flat.grad - g).norm() / g.norm().clamp_min(1e-30)) for o, g in zip(optimizers, local_a)]
        line = f"rank {rank} trial {t}: loss a/b/c {la:.7f} {lb:.7f} {lc:.7f}  ddp-vs-allreduce {['%.1e' % e for e in errs]}  bare-vs-bare {['%.1e' % e for e in rep]}"
        for k, (o, w) in enumerate(zip(optimizers, want)):
            if errs[k] > 1e-5 or rep[k] > 1e-5:
                ref = w if errs[k] > 1e-5 else local_a[k]
                d = (o.flat.grad - ref).abs() if errs[k] <= 1e-5 else None
                line += f"\n   arena {k}: {len(o.flat.params)} parameters"
        print(line, flush=True)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, trials), nprocs=2, join=True)
