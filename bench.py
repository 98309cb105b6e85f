#!/usr/bin/env python3
"""Benchmark of the FovealSeg hot path on MI355X.

metric  : images/sec, forward + backward + optimiser step, 1024x1024 -> 80x80 foveated HRNetV2 + C1,
          batch 64 per GPU (BASELINE.json configs[1]), train mode (BN batch statistics, Dropout 0.3).
step    : one full optimisation step (train_deform_semantic.py:74-129) on a synthetic batch that is
          already resident in HBM.
Usage   : python bench.py --gpus N --steps K --warmup W        (N>1 under torch.distributed.run)
Output  : ONE JSON line on rank 0 (see README/DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, dense bf16 MFMA; the bf16x3 conv issues 6 bf16 MFMAs per fp32 product
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16 MFMA (same rate as bf16); the f16x2 conv issues 3 fp16 MFMAs per fp32 product
PEAK_HBM_GBS = 8000.0


def _profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes over this same command
    (profiles/r01/hbm_traffic_serial.json; FETCH_SIZE doubled per the gfx950 correction).  PMC counters
    cannot be read from inside the benchmark process, so this is the offline measurement, or None."""
    path = os.path.join(ROOT, "profiles", "r01", "hbm_traffic_serial.json")
    try:
        with open(path) as f:
            table = json.load(f)
        hits = [v for k, v in table.items() if k == kernel or k.startswith(kernel + "<")]      # template instances of one kernel
        n = sum(v["launches"] for v in hits)
        avg = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n
        return {"hbm_bytes_per_launch": int(avg), "source": "profiles/r01/hbm_traffic_serial.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    except Exception:
        return None


def cpu_baseline(batch=4, H=1024, threads=None, reps=5):
    """The CPU oracle (port of the reference algorithm) timed on this host: BASELINE.json configs[0] (B=4, 1024^2), one
    warm-up and five timed train-mode forward+backward passes, median reported (about 15 s of CPU work)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fovealseg_oracle as O
    from fovealseg.weights import apply_name_keyed_init
    from fovealseg.train import synthetic_batch
    if threads is None:          # the GPU box exposes 128 logical CPUs but grants a 16-core share
        try:
            threads = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    o = O.OracleDeformSeg()
    apply_name_keyed_init(o)
    o.train()
    X, Fp, Y, cls = synthetic_batch(batch, H, H, seed=1, device="cpu")
    times = []
    for it in range(1 + reps):
        feed = {"img_data": X, "seg_label": Y.clone(), "focus_point": Fp, "cls_label": cls}
        o.zero_grad()
        t0 = time.perf_counter()
        loss, acc, edge = o(feed)
        loss.backward()
        times.append(time.perf_counter() - t0)
    timed = sorted(times[1:])
    t = timed[len(timed) // 2]
    return {"value": round(batch / t, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/fovealseg_oracle.py OracleDeformSeg, B={batch}, {H}x{H}->80x80, train mode, 1 warm-up + {reps} timed fwd+bwd "
                      f"(no optimiser step), median {t:.2f} s, {sum(times):.1f} s of CPU work in all"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE config 2: 64)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--serial-streams", action="store_true", help="run the HRNet branches on one stream (profiling)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--conv-precision", default="f16x2", choices=["f16x2", "bf16x3", "f32"],
                    help="arithmetic of the conv kernels (fp32 tensors and fp32 accumulation in every mode): f16x2 = operands "
                         "scaled and split into 2 fp16 terms, 3 fp16 MFMAs per product; "
                         "bf16x3 = 3 bf16 terms, 6 bf16 MFMAs per product; f32 = fp32 MFMA.  All three are at the fp32 error level "
                         "(tools/conv_accuracy.py)")
    args = ap.parse_args()

    import fovealseg
    from fovealseg import train as T
    from fovealseg import ops

    fovealseg.hip.load()       # fail loudly if the HIP library is missing
    fovealseg.hip.set_conv_precision(args.conv_precision)
    rank, local_rank, world = T.ddp_setup(backend=args.backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    T.broadcast_parameters(optimizers, module)
    batch = T.synthetic_batch(args.batch, args.size, args.size, seed=1 + rank, device=dev)
    ops.DropoutState.seed = 1234 + rank

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from fovealseg import modules as Mods
    if args.serial_streams:
        Mods.PARALLEL_BRANCHES = False
    out = None
    for i in range(args.warmup):
        out = T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    loss_val = float(out[0].detach())
    assert loss_val == loss_val, "loss is NaN"

    # Per-kernel roofline pass.  In the timed region the HRNet branches run on 4 HIP streams, so
    # kernel lifetimes overlap and a per-launch duration is not separable; the same K steps are
    # therefore repeated with the branch streams serialised and every conv launch bracketed by HIP
    # events on the launch stream (this pass is NOT part of `value`).
    # forward-only (inference) rate of the same path, SURVEY.md 8(d): eval mode (running statistics, no dropout), no autograd
    fwd_only = None
    if world == 1:
        try:
            module.eval()
            T.eval_step(module, batch)
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for i in range(args.steps):
                T.eval_step(module, batch)
            torch.cuda.synchronize()
            tf = time.perf_counter() - tf0
            fwd_only = {"img_per_s": round(args.batch * args.steps / tf, 1), "ms_per_batch": round(1000.0 * tf / args.steps, 2),
                        "algorithmic_tflops": round(0.15229 * args.batch * args.steps / tf, 1),
                        "what": "DeformSegmentationModule forward, is_inference=True, eval mode, no_grad; 152.29 GFLOP per image"}
        except Exception as exc:
            fwd_only = {"error": repr(exc)}
        finally:
            module.train()

    timer, timer_error = None, None
    if not args.no_kernel_timer and world == 1:      # single-GPU only: the step contains collectives
        saved = Mods.PARALLEL_BRANCHES
        Mods.PARALLEL_BRANCHES = False
        timer = ops.KernelTimer()
        ops.TIMER = timer
        try:
            torch.cuda.synchronize()
            tr0 = time.perf_counter()
            for i in range(args.steps):
                T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=args.warmup + args.steps + i)
            torch.cuda.synchronize()
            serial_elapsed = time.perf_counter() - tr0
        except Exception as exc:                       # the throughput line must survive a failure of the diagnostic pass
            timer, timer_error = None, repr(exc)
        finally:
            ops.TIMER = None
            Mods.PARALLEL_BRANCHES = saved
    if world > 1:
        dist.barrier()

    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    imgs = args.batch * world * args.steps
    line = {
        "metric": "images/sec fwd+bwd, 1024->80 foveated HRNetV2, batch 64",
        "value": round(imgs / elapsed, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "conv_precision": args.conv_precision,
        "config": {"workload": f"BASELINE configs[1]: HRNetV2-nodownsp + C1, {args.size}x{args.size}->80x80, gaussian_radius 45, "
                               f"batch {args.batch}/GPU, train mode (BN batch stats, Dropout 0.3), fwd+bwd+Adam x4",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}", "loss": round(loss_val, 5)},
    }
    if rank == 0:
        if timer_error is not None:
            line["roofline"] = None
            line["roofline_error"] = timer_error
        if timer is not None:
            summ = timer.summary()
            def entry(kk, desc, peak, kname):
                ach = kk["flops"] / (kk["total_ms"] * 1e-3) / 1e12
                return {"kernel": desc, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": _profiled_traffic(kname),
                        "launches_per_step": kk["launches"] // args.steps,
                        "avg_launch_us": round(1000.0 * kk["total_ms"] / kk["launches"], 2),
                        "gflop_per_launch": round(kk["flops"] / kk["launches"] / 1e9, 3),
                        "share_of_serial_step": round(kk["total_ms"] / (1000.0 * serial_elapsed), 3)}
            split = {"f16x2": ("PrecF16", PEAK_F16_MFMA_TFLOPS / 3.0, "two scaled fp16 terms per operand, 3 x v_mfma_f32_32x32x16_f16 per product"),
                     "bf16x3": ("PrecX3", PEAK_BF16_MFMA_TFLOPS / 6.0, "three bf16 terms per operand, 6 x v_mfma_f32_32x32x16_bf16 per product")}
            if args.conv_precision in split and "conv3x3" in summ:
                tag, peak, how = split[args.conv_precision]
                line["roofline"] = entry(
                    summ["conv3x3"],
                    f"conv3x3_halo_kernel<{tag}> (3x3 stride-1 forward + bwd-data, halo-tiled implicit GEMM; {how}, fp32 accumulate; "
                    "achieved = algorithmic fp32 FLOP/s over the C-ABI call incl. its weight pack pre-kernels, peak = dense MFMA peak / "
                    "MFMAs per product)", peak, "conv3x3_halo_kernel")
                line["roofline"]["serial_ms_per_step"] = round(1000.0 * serial_elapsed / args.steps, 2)
                line["roofline"]["measured"] = "second pass of the same K steps with branch streams serialised, HIP events per launch"
                if "wgrad3x3" in summ:
                    line["roofline_wgrad"] = entry(
                        summ["wgrad3x3"], f"conv_wgrad_class_kernel<{tag},3,3> (3x3 stride-1 bwd-weight, 9 taps per workgroup, split-K atomics; {how})",
                        peak, f"conv_wgrad_class_kernel<fs_split::{tag}, 3, 3>")
                if "conv_affine" in summ:
                    line["roofline_other_convs"] = entry(
                        summ["conv_affine"], f"conv_tapset_kernel<{tag}> (strided 3x3 forward + bwd-data sub-problems) + conv_igemm_split_kernel<{tag}> (1x1 convs "
                                             "and single-tap sub-problems, HBM-bound at these channel counts)",
                        peak, f"conv_igemm_split_kernel<fs_split::{tag}>")
            elif "conv_affine" in summ:
                line["roofline"] = entry(summ["conv_affine"], "conv_igemm_affine_kernel<1> (fwd + bwd-data implicit GEMM, fp32 MFMA 32x32x2)",
                                         PEAK_F32_MFMA_TFLOPS, "conv_igemm_affine_kernel<1>")
                line["roofline"]["serial_ms_per_step"] = round(1000.0 * serial_elapsed / args.steps, 2)
                if "conv_wgrad" in summ:
                    line["roofline_wgrad"] = entry(summ["conv_wgrad"], "conv_wgrad_taps_kernel<3|9> + conv_wgrad_kernel (bwd-weight, fp32 MFMA)",
                                                   PEAK_F32_MFMA_TFLOPS, "conv_wgrad_taps_kernel")
            def hbm_entry(kk, desc, knames):
                ach = kk["flops"] / (kk["total_ms"] * 1e-3) / 1e9
                parts = [_profiled_traffic(k) for k in knames]
                traffic = None
                if all(parts):
                    traffic = {"hbm_bytes_per_launch": sum(t["hbm_bytes_per_launch"] for t in parts), "source": parts[0]["source"]}
                return {"kernel": desc, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                        "launches_per_step": kk["launches"] // args.steps,
                        "avg_launch_us": round(1000.0 * kk["total_ms"] / kk["launches"], 2),
                        "mb_per_launch": round(kk["flops"] / kk["launches"] / 1e6, 2),
                        "share_of_serial_step": round(kk["total_ms"] / (1000.0 * serial_elapsed), 3)}
            if "bn_fwd" in summ:
                line["roofline_bn_fwd"] = hbm_entry(summ["bn_fwd"], "bn_act_fwd_kernel (normalise + residual + activation + mask bytes; "
                                                    "algorithmic bytes = conv output read, [residual read,] activation written, 1 mask byte per 4 channels)", ["bn_act_fwd_kernel"])
            if "bn_bwd" in summ:
                line["roofline_bn_bwd"] = hbm_entry(summ["bn_bwd"], "bn_bwd_reduce_kernel + bn_bwd_apply_kernel (one C-ABI call; algorithmic bytes = dz, conv "
                                                    "output and mask read once, dy [and the residual gradient] written once -- the two-pass "
                                                    "reduction reads dz and the conv output twice, so 3/5 of the HBM peak is this pair's ceiling)",
                                                    ["bn_bwd_reduce_kernel", "bn_bwd_apply_kernel"])
        if fwd_only is not None:
            line["forward_only"] = fwd_only
        # whole step against the planning roofs of SURVEY.md 8(d): 456.9 GFLOP and 3 x 0.671 GB algorithmic per image, fwd+bwd
        per_gpu = line["value"] / world
        line["whole_step"] = {"algorithmic_tflops": round(0.4569 * per_gpu, 1), "fp32_mfma_peak_tflops": PEAK_F32_MFMA_TFLOPS,
                              "frac_of_fp32_mfma_peak": round(0.4569 * per_gpu / PEAK_F32_MFMA_TFLOPS, 3),
                              "algorithmic_hbm_gbs": round(3 * 0.671 * per_gpu, 1),
                              "note": "per GPU; the split-precision modes run the products on the 16-bit MFMA pipes, so the fp32-MFMA roof (340 img/s) is not their ceiling"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as exc:
                line["cpu_baseline"] = None
                line["cpu_baseline_error"] = repr(exc)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
