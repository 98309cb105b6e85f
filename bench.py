#!/usr/bin/env python3
"""Benchmark of the FovealSeg hot path on MI355X.

metric  : images/sec, forward + backward + optimiser step, 1024x1024 -> 80x80 foveated HRNetV2 + C1,
          batch 64 per GPU (BASELINE.json configs[1]), train mode (BN batch statistics, Dropout 0.3).
step    : one full optimisation step (train_deform_semantic.py:74-129) on a synthetic batch that is
          already resident in HBM.
modes   : the conv engine has three arithmetic modes, all fp32 tensors + fp32 accumulation.
          All three are timed in ONE run.  `value` is the `--headline` mode, by default `bf16x3`: operands
          split into three bf16 terms = 24 significand bits, the reference's fp32 operand width.  `f16x2`
          (22-23 bit operands) and `f32` (fp32 MFMA) are reported next to it under `modes`.
Usage   : python bench.py --gpus N --steps K --warmup W
          N>1: under torch.distributed.run (one rank per GPU, RCCL); started WITHOUT a launcher it
          spawns that launcher itself before touching the GPU.
Output  : ONE JSON line on rank 0 (see README/DESIGN.md "Measurement").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
T_START = time.perf_counter()      # (main() resets it: wall seconds of the phases go into the line as "wall")
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # same guide, dense bf16 / fp16 MFMA
PEAK_HBM_GBS = 8000.0

MODES = {
    # name: (kernel template tag, MFMAs per fp32 product, operand significand bits, description)
    "bf16x3": ("PrecX3", 6, 24, "f32 storage + f32 accumulate; each operand = 3 bf16 terms (24 significand bits), 6 x v_mfma_f32_32x32x16_bf16 per product"),
    "f16x2": ("PrecF16", 3, 22, "f32 storage + f32 accumulate; each operand scaled and split into 2 fp16 terms (22-23 significand bits), 3 x v_mfma_f32_32x32x16_f16 per product"),
    "f32": (None, 1, 24, "f32 storage, v_mfma_f32_32x32x2_f32 products, f32 accumulate"),
}


def mode_peak(mode):
    return PEAK_F32_MFMA_TFLOPS if mode == "f32" else PEAK_16BIT_MFMA_TFLOPS / MODES[mode][1]


def _profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes over this same command
    (profiles/rNN/hbm_traffic_serial*.json; FETCH_SIZE doubled per the gfx950 correction).  PMC counters
    cannot be read from inside the benchmark process, so this is the offline measurement, or None."""
    for rel in ("profiles/r05/hbm_traffic_serial.json", "profiles/r04/hbm_traffic_serial.json", "profiles/r03/hbm_traffic_serial.json", "profiles/r02/hbm_traffic_serial.json", "profiles/r01/hbm_traffic_serial.json"):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                table = json.load(f)
            # all template instances of one kernel: "name" matches "name<...>", "name<a, b" matches "name<a, b, c>"
            names = kernel if isinstance(kernel, (list, tuple)) else [kernel]
            hits = [v for k, v in table.items() if any(k == n or k.startswith(n + "<") or ("<" in n and k.startswith(n)) for n in names)]
            n = sum(v["launches"] for v in hits)
            avg = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits) / n
            return {"hbm_bytes_per_launch": int(avg), "source": rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
        except Exception:
            continue
    return None


def _profiled_traffic_table():
    for rel in ("profiles/r05/hbm_traffic_serial.json", "profiles/r04/hbm_traffic_serial.json", "profiles/r03/hbm_traffic_serial.json"):
        try:
            with open(os.path.join(ROOT, rel)) as f:
                return json.load(f), rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except Exception:
            continue
    return None


def _profiled_mfma_busy(mode, kernel):
    """MFMA-pipe busy fraction and sustained clock of the dominant kernel from the committed SQ-counter passes
    (profiles/r02/pmc/sq_<mode>_<kernel>_fwd_shape0.txt: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)), or None."""
    import ast
    import re
    rel = f"profiles/r05/pmc/sq_{mode}_{kernel}_fwd_shape0.txt"
    for older in ("r04", "r03", "r02"):
        if not os.path.exists(os.path.join(ROOT, rel)):
            rel = f"profiles/{older}/pmc/sq_{mode}_{kernel}_fwd_shape0.txt"
    try:
        vals = {}
        for line in open(os.path.join(ROOT, rel)):
            m = re.search(r"(\{.*\})", line)
            if m:
                vals.update(ast.literal_eval(m.group(1)))
        per_xcd = vals["GRBM_GUI_ACTIVE"] / 8.0
        return {"mfma_pipe_busy": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (per_xcd * 1024.0), 3),
                "sustained_clock_ghz": round(per_xcd / (vals["_dur"] * 1e3), 2), "shape": "64->64 3x3 @ 80x80, B=64", "source": rel}
    except Exception:
        return None


def _profiled_kernel_avg_us(knames):
    """Average duration of the named kernels in the committed `rocprofv3 --kernel-trace --stats` summary of this same command with the
    branch streams serialised (profiles/r05/bench_serial_kernel_stats_bf16x3.csv): the KERNEL alone, where `avg_launch_us` of the line is
    an in-process event bracket around the C-ABI call including its weight-pack pre-kernel."""
    import csv
    for rel in ("profiles/r05/bench_serial_kernel_stats_bf16x3.csv", "profiles/r04/bench_serial_kernel_stats_bf16x3_final_build.csv"):
        try:
            tot, calls = 0.0, 0
            with open(os.path.join(ROOT, rel)) as f:
                for row in csv.DictReader(f):
                    if any(k + "<" in row["Name"] for k in knames):
                        tot += float(row["TotalDurationNs"]); calls += int(row["Calls"])
            if calls:
                return {"kernel_avg_us": round(tot / calls / 1e3, 2), "source": rel}
        except Exception:
            continue
    return None


def takes_f43(W, Cs, Cd):
    """csrc/conv_wino.hip: wino4_selected -- which 3x3 stride-1 problems of the bf16x3 mode run on the F(4,3) kernel (the rest: F(2,3))."""
    return not (os.environ.get("FS_WINO4", "1") == "0" or W % 4 or W < 8)


def _threads():
    try:          # the GPU box exposes 128 logical CPUs but grants a 16-core share
        return min(16, len(os.sched_getaffinity(0)))
    except AttributeError:
        return min(16, os.cpu_count() or 1)


def cpu_baseline(batch=4, H=1024, threads=None, reps=5, fwd_batch=64):
    """The CPU oracle (port of the reference algorithm) timed on this host: BASELINE.json configs[0] (B=4, 1024^2), one
    warm-up and five timed train-mode forward+backward passes, median reported, plus one B=64 forward-only pass
    (about 30 s of CPU work in all)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fovealseg_oracle as O
    from fovealseg.weights import apply_name_keyed_init
    from fovealseg.train import synthetic_batch
    threads = threads or _threads()
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
    out = {"value": round(batch / t, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"oracle/fovealseg_oracle.py OracleDeformSeg, B={batch}, {H}x{H}->80x80, train mode, 1 warm-up + {reps} timed fwd+bwd "
                     f"(no optimiser step), median {t:.2f} s, {sum(times):.1f} s of CPU work in all",
           # what the one-line record carries (the full sentence above goes to bench_detail.json)
           "sample_short": f"oracle OracleDeformSeg B={batch} {H}x{H}->80x80 train fwd+bwd: 1 warm-up + {reps} timed, median {t:.2f} s, "
                           f"{sum(times):.0f} s CPU in all"}
    if fwd_batch:
        o.eval()
        X, Fp, Y, cls = synthetic_batch(fwd_batch, H, H, seed=1, device="cpu")
        with torch.no_grad():
            t0 = time.perf_counter()
            o({"img_data": X, "seg_label": Y, "focus_point": Fp, "cls_label": cls}, is_inference=True)
            tf = time.perf_counter() - t0
        out["forward_only"] = {"value": round(fwd_batch / tf, 3), "unit": "img/s", "sample": f"one eval-mode forward, B={fwd_batch}, {tf:.1f} s"}
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`bench.py --gpus N` (N>1) started without a launcher: become the launcher.  Nothing in this process has touched the
    GPU yet (torch is not even imported), so the N ranks are plain children: `python -m torch.distributed.run ... bench.py`
    -- the command the driver itself uses (train_deform_semantic.py:687-689 uses mp.spawn for the same purpose)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE config 2: 64)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive (host-fed) pass recorded in bench_detail.json")
    ap.add_argument("--serial-streams", action="store_true", help="run the HRNet branches on one stream (profiling)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--headline", default="bf16x3", choices=list(MODES),
                    help="the mode whose throughput is `value` (default bf16x3: 24-bit operands like the reference's fp32)")
    ap.add_argument("--modes", default="bf16x3,f16x2,f32", help="comma list of conv arithmetic modes to time in this run")
    ap.add_argument("--conv-precision", default=None, choices=list(MODES), help="shorthand: time ONLY this mode and make it the headline")
    return ap.parse_args()


def main():
    global T_START
    T_START = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    import fovealseg
    from fovealseg import train as T
    from fovealseg import ops
    from fovealseg import modules as Mods

    if args.conv_precision:
        args.headline, args.modes = args.conv_precision, args.conv_precision
    modes = [m for m in args.modes.split(",") if m]
    if args.headline not in modes:
        modes.insert(0, args.headline)
    modes.sort(key=lambda m: m != args.headline)          # headline first

    fovealseg.hip.load()       # fail loudly if the HIP library is missing
    rank, local_rank, world = T.ddp_setup(backend=args.backend)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == world
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    cfg = fovealseg.lvis50_cfg()
    module, nets = T.build_module(cfg, device=dev)
    module.train()
    optimizers = T.create_optimizers(nets, cfg)
    T.broadcast_parameters(optimizers, module)
    batch = T.synthetic_batch(args.batch, args.size, args.size, seed=1 + rank, device=dev)
    ops.DropoutState.seed = 1234 + rank
    if args.serial_streams:
        Mods.PARALLEL_BRANCHES = False

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = [0]

    def steps(n):
        out = None
        for _ in range(n):
            out = T.train_step(module, optimizers, batch, cfg, epoch=1, cur_iter=it[0])
            it[0] += 1
        return out

    def max_over_ranks(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    results = {}
    comm = {}
    # wall seconds of each phase of this process (a GPU-busy average over the whole run mixes the timed passes with the CPU
    # baseline): which seconds the GPU was the one working, and which the host cores
    phases = {"setup_s": round(time.perf_counter() - T_START, 1)}
    tp = time.perf_counter()
    for mode in modes:
        fovealseg.hip.set_conv_precision(mode)
        for opt in optimizers:
            opt.flat.refresh_amax()
        steps(args.warmup)
        barrier()
        T.COMM_EVENTS = [] if world > 1 else None        # HIP events around the four arena all-reduces of every timed step (negligible cost)
        t0 = time.perf_counter()
        out = steps(args.steps)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        if T.COMM_EVENTS:
            ms = [a.elapsed_time(b) for a, b in T.COMM_EVENTS]
            comm[mode] = {"allreduce_ms_per_step": round(sum(ms) / len(ms), 3), "allreduce_ms_max": round(max(ms), 3), "exchanges": len(ms)}
        T.COMM_EVENTS = None
        loss_val = float(out[0].detach())
        assert loss_val == loss_val, f"loss is NaN in mode {mode}"
        res = {"value": round(args.batch * world * args.steps / elapsed, 3), "unit": "img/s", "ms_per_step": round(1000.0 * elapsed / args.steps, 3),
               "operand_significand_bits": MODES[mode][2], "arithmetic": MODES[mode][3], "loss": round(loss_val, 5)}

        # Per-kernel roofline pass.  In the timed region the HRNet branches run on 4 HIP streams, so kernel lifetimes overlap and
        # a per-launch duration is not separable; the same K steps are therefore repeated with the branch streams serialised and
        # every conv / BatchNorm / front-end launch bracketed by HIP events on the launch stream (NOT part of `value`).
        # (All ranks run it -- train_step holds the gradient all-reduce -- and rank 0's kernels are the ones reported.)
        if not args.no_kernel_timer:
            saved = Mods.PARALLEL_BRANCHES
            Mods.PARALLEL_BRANCHES = False
            timer = ops.KernelTimer()
            ops.TIMER = timer
            try:
                barrier()
                tr0 = time.perf_counter()
                steps(args.steps)
                barrier()
                serial_elapsed = time.perf_counter() - tr0
                res.update(roofline_entries(mode, timer.summary(), args.steps, serial_elapsed, timer))
                res.update(family_entries(mode, timer, args.steps))
            except Exception as exc:                       # the throughput line must survive a failure of the diagnostic pass
                res["roofline"] = None
                res["roofline_error"] = repr(exc)
            finally:
                ops.TIMER = None
                Mods.PARALLEL_BRANCHES = saved
        per_gpu = res["value"] / world
        res["whole_step"] = {"algorithmic_tflops": round(0.4569 * per_gpu, 1), "frac_of_mode_mfma_peak": round(0.4569 * per_gpu / mode_peak(mode), 4),
                             "mode_mfma_peak_tflops": round(mode_peak(mode), 1), "algorithmic_hbm_gbs": round(3 * 0.671 * per_gpu, 1)}
        results[mode] = res
        if world > 1:
            dist.barrier()

    phases["gpu_timed_and_kernel_timer_passes_s"] = round(time.perf_counter() - tp, 1)
    tp = time.perf_counter()
    # forward-only (inference) rate of the same path, SURVEY.md 8(d): eval mode (running statistics, no dropout), no autograd
    fwd_only = None
    if world == 1 and not args.no_forward_only:
        fwd_only = {}
        for mode in modes:
            fovealseg.hip.set_conv_precision(mode)
            try:
                module.eval()
                T.eval_step(module, batch)
                torch.cuda.synchronize()
                tf0 = time.perf_counter()
                for _ in range(args.steps):
                    T.eval_step(module, batch)
                torch.cuda.synchronize()
                tf = time.perf_counter() - tf0
                fwd_only[mode] = {"img_per_s": round(args.batch * args.steps / tf, 1), "ms_per_batch": round(1000.0 * tf / args.steps, 2),
                                  "algorithmic_tflops": round(0.15229 * args.batch * args.steps / tf, 1)}
            except Exception as exc:
                fwd_only[mode] = {"error": repr(exc)}
            finally:
                module.train()
        fwd_only["what"] = "DeformSegmentationModule forward, is_inference=True, eval mode, no_grad; 152.29 GFLOP per image"

    # PCIe-inclusive rate (SURVEY 8(d) asks for both; never `value`): every step consumes a NEW batch that starts in pinned host memory as
    # decoded uint8 samples and reaches the device through data.DevicePrefetcher (copy stream + fs_ingest_sample one batch ahead)
    h2d = None
    if world == 1 and not args.no_h2d:
        try:
            h2d = host_fed_rate(T, module, optimizers, cfg, args, dev, modes[0], results[modes[0]]["value"])
        except Exception as exc:
            h2d = {"error": repr(exc)}
    phases["gpu_forward_only_and_host_fed_s"] = round(time.perf_counter() - tp, 1)
    distributed = {"world_size": world, "backend": dist.get_backend() if dist.is_initialized() else None,
                   "rccl_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                   "gradient_bytes_per_step": int(sum(o.flat.grad.numel() for o in optimizers) * 4), "comm": comm or None,
                   "note": "one SUM all-reduce per flat gradient arena after the backward (4 per step), 1/world folded into Adam; "
                           "allreduce_ms = HIP events around the four collectives on rank 0"}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            tp = time.perf_counter()
            try:
                cpu = cpu_baseline()
            except Exception as exc:
                cpu = {"error": repr(exc)}
            phases["cpu_baseline_gpu_idle_s"] = round(time.perf_counter() - tp, 1)
        line, detail = build_lines(args.headline, world, args.steps, args.warmup, args.batch, args.size, results, fwd_only, cpu)
        detail["h2d_included"] = h2d
        detail["phases"] = phases
        line["wall"] = {"gpu_s": round(phases["gpu_timed_and_kernel_timer_passes_s"] + phases["gpu_forward_only_and_host_fed_s"], 1),
                        "cpu_baseline_s": phases.get("cpu_baseline_gpu_idle_s", 0.0), "setup_s": phases["setup_s"]}
        assert len(json.dumps(line)) < MAX_LINE_BYTES
        detail["distributed"] = distributed
        write_detail(detail)
        print(json.dumps(line), flush=True)          # the LAST thing on stdout, < 2 KB: the driver parses this line
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_fed_rate(T, module, optimizers, cfg, args, dev, mode, resident_value):
    """img/s of the same training step when every batch crosses PCIe inside the timed region (tools/host_fed_bench.py, in the bench)."""
    import numpy as np
    import torch
    import fovealseg
    from fovealseg import data
    fovealseg.hip.set_conv_precision(mode)
    B, H = args.batch, args.size
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:H, 0:H]
    disk = (((yy - H // 2) ** 2 + (xx - H // 2) ** 2) <= (0.15 * H) ** 2).astype(np.uint8)
    host_batches = []
    for _ in range(2):            # two distinct host batches, alternated; pinned once, as a loader's pin_memory thread delivers them
        samples = []
        for _i in range(B):
            img = torch.from_numpy(rng.integers(0, 256, (H, H, 4), dtype=np.uint8)).pin_memory()
            dy, dx = (int(v) for v in rng.integers(-H // 4, H // 4, 2))
            mask = torch.from_numpy(np.roll(disk, (dy, dx), (0, 1))).pin_memory()
            samples.append(data.Sample(img, mask, (0, 0, 0, 0), (H // 2 + dy, H // 2 + dx), (H, H), int(rng.integers(0, 50))))
        host_batches.append(samples)
    n_warm, n = 2, args.steps
    it = data.DevicePrefetcher((host_batches[i % 2] for i in range(n_warm + n)), dev, channels=4)
    k = 0
    t0 = None
    for bt in it:
        if k == n_warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        T.train_step(module, optimizers, bt, cfg, epoch=1, cur_iter=1000 + k)
        k += 1
    torch.cuda.synchronize()
    rate = B * n / (time.perf_counter() - t0)
    return {"value": round(rate, 2), "unit": "img/s", "conv_precision": mode, "steps": n, "ratio_to_resident": round(rate / resident_value, 4),
            "h2d_mb_per_batch": round(B * H * H * 5 / 1e6, 1),
            "what": "same step, a new batch per step from pinned host memory: uint8 RGBA image + uint8 mask per sample over PCIe on a copy "
                    "stream, fs_ingest_sample (ToTensor, padding, batch assembly) on the device one batch ahead (data.DevicePrefetcher); "
                    "the reference uploads fp32 (4x the bytes) through DDP's scatter (train_deform_semantic.py:74-95)"}


MAX_LINE_BYTES = 2048


def _short_roofline(r):
    """The one roofline object of the JSON line: fixed keys, kernel name <= 80 characters."""
    if not r:
        return None
    keys = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_measured_in_this_run", "launches_per_step", "avg_launch_us", "kernel_avg_us",
            "executed_mfma_fraction")
    out = {"kernel": str(r.get("kernel", "")).split(" (")[0][:80]}
    out.update({k: r.get(k) for k in keys})
    return out


def build_lines(headline, world, steps, warmup, batch, size, results, fwd_only, cpu):
    """(line, detail): `line` is the ONE JSON object printed last on stdout (bench contract; kept under MAX_LINE_BYTES so
    the driver's stdout tail always holds all of it), `detail` everything else (per-mode rooflines of every kernel class,
    front-end table, forward-only rates, long descriptions) for bench_detail.json."""
    head = results[headline]
    line = {
        "metric": "images/sec fwd+bwd, 1024->80 foveated HRNetV2, batch 64",
        "value": head["value"], "unit": "img/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "conv_precision": headline,
        "config": {"workload": f"configs[1]: HRNetV2-nodownsp+C1 {size}x{size}->80x80 r45 B={batch}/GPU train fwd+bwd+Adam",
                   "global_batch": batch * world, "parallelism": f"dp{world}"},
        "roofline": _short_roofline(head.get("roofline")),
    }
    if cpu is not None:
        line["cpu_baseline"] = {k: v for k, v in cpu.items() if k in ("value", "unit", "cores", "kind", "error")}
        if "sample" in cpu:
            line["cpu_baseline"]["sample"] = str(cpu.get("sample_short", cpu["sample"]))[:160]
    line["modes"] = {m: {"value": r["value"], "ms_per_step": r["ms_per_step"],
                         "frac": (r.get("roofline") or {}).get("frac")} for m, r in results.items()}
    line["detail"] = "bench_detail.json"
    detail = {"line": line, "modes": results, "forward_only": fwd_only, "cpu_baseline": cpu}
    assert len(json.dumps(line)) < MAX_LINE_BYTES, "bench line grew past what the driver keeps"
    return line, detail


def write_detail(detail):
    """Side file with everything the line leaves out (next to bench.py, and under gpurun_out/ so it travels back from the GPU box)."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    json.dump(detail, f, indent=1)
        except OSError:
            pass


# Front-end / loss / optimiser kernels (HBM or gather-latency bound): timer kind -> description
FRONTEND_KINDS = {
    # timer kind: (kernel name in the rocprofv3 tables, description)
    "fe_gaze_lowres": ("gaze_lowres_kernel", "K1: bilinear 1024^2 -> 80^2 taps + gaze map (gather: 4 taps per output)"),
    "fe_area_pool": ("area_pool_kernel", "K3: label 1024^2 -> 80^2 area pooling, the only full-resolution pass"),
    "fe_gauss_grid_fwd": ("gauss_grid_fwd_kernel", "K4: separable 91-tap Gaussian saliency accumulation -> grid (arithmetic-bound: 3 x 2 x 91 fp64 FMAs per grid point)"),
    "fe_gauss_grid_bwd": ("gauss_grid_bwd_kernel", "K4 backward"),
    "fe_grid_sample_fwd": ("grid_sample_fwd_kernel", "K5: foveated bilinear gather of the image (4 taps x 3 planes per output, one 64-B sector each)"),
    "fe_grid_sample_label": ("grid_sample_label_kernel", "K5: label gather + truncation"),
    "fe_grid_sample_bwd_grid": ("grid_sample_bwd_grid_kernel", "K6: d loss / d grid"),
    "fe_seg_loss_fwd": ("seg_loss_fwd_kernel", "K10/K11: Focal + Dice + accuracies over pred (3 passes over 51 classes per pixel: exp/log-bound)"),
    "fe_seg_loss_bwd": ("seg_loss_bwd_kernel", "K10/K11 backward"),
    "fe_adam": ("adam_kernel", "one launch per arena; 16 B read + 12 B written per parameter"),
}


def roofline_entries(mode, summ, nsteps, serial_elapsed, timer=None):
    tag, n_mfma, _bits, how = MODES[mode]
    peak = mode_peak(mode)
    out = {}

    def entry(kk, desc, kname):
        ach = kk["flops"] / (kk["total_ms"] * 1e-3) / 1e12
        tr = _profiled_traffic(kname)
        return {"kernel": desc, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": tr["hbm_bytes_per_launch"] if tr else None, "traffic_unit": "HBM bytes per launch",
                "traffic_source": tr["source"] if tr else None, "traffic_measured_in_this_run": False,
                "launches_per_step": kk["launches"] // nsteps,
                "avg_launch_us": round(1000.0 * kk["total_ms"] / kk["launches"], 2),
                "gflop_per_launch": round(kk["flops"] / kk["launches"] / 1e9, 3),
                "share_of_serial_step": round(kk["total_ms"] / (1000.0 * serial_elapsed), 3)}

    def hbm_entry(kk, desc, knames, per="launch"):
        ach = kk["flops"] / (kk["total_ms"] * 1e-3) / 1e9
        traffic, source = None, None
        if per == "layer":
            # bytes of ALL the named kernels per launch of the LAST one (the pass every layer runs): kernels that only some layers run
            # (the reduction pass where no producer formed the sums) count with their share
            tr = _profiled_traffic_table()
            if tr is not None:
                table, source = tr
                rows = [[v for k, v in table.items() if k == n or k.startswith(n + "<")] for n in knames]
                if rows[-1]:
                    base = sum(v["launches"] for v in rows[-1])
                    traffic = int(sum(v["hbm_bytes_per_launch"] * v["launches"] for r in rows for v in r) / base)
        else:
            parts = [_profiled_traffic(k) for k in knames]
            if all(parts):
                traffic, source = sum(t["hbm_bytes_per_launch"] for t in parts), parts[0]["source"]
        return {"kernel": desc, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": source,
                "launches_per_step": kk["launches"] // nsteps,
                "avg_launch_us": round(1000.0 * kk["total_ms"] / kk["launches"], 2),
                "mb_per_launch": round(kk["flops"] / kk["launches"] / 1e6, 2),
                "share_of_serial_step": round(kk["total_ms"] / (1000.0 * serial_elapsed), 3)}

    if tag is not None and "conv3x3" in summ:
        # every 3x3 stride-1 layer of this workload has an even width.  bf16x3: the layers whose width is a multiple of 4 take the F(4,3) row
        # kernel of round 5 (csrc/conv_wino4.hip; takes_f43 above), the 10-wide maps and the wide 20-wide layers the F(2,3) kernels; f16x2: the
        # layers with >= 128 source channels take F(2,3) (csrc/conv_wino.hip: fs_wino_eligible), the rest the plain halo kernel
        wino = os.environ.get("FS_WINOGRAD", "1") != "0"
        f43 = wino and mode == "bf16x3" and os.environ.get("FS_WINO4", "1") != "0"
        kname = (["conv3x3_wino4_kernel", "conv3x3_wino48_kernel"] if f43 else []) + (["conv3x3_wino_kernel", "conv3x3_wino8_kernel"] if wino else ["conv3x3_halo_kernel"])
        if f43:
            what = (f"conv3x3_wino4_kernel<{tag}> (+ its eight-wave form conv3x3_wino48_kernel on wide layers, conv3x3_wino8_kernel on the 10-wide maps; 3x3 stride-1 "
                    "forward + bwd-data, halo-tiled implicit GEMM with F(4,3) minimal filtering along the row: 18 matrix steps per four output pixels where the "
                    f"direct form spends 36 and F(2,3) 24; {how}; ")
        else:
            what = (f"conv3x3_wino_kernel<{tag}> (+ its eight-wave form conv3x3_wino8_kernel on wide layers; 3x3 stride-1 forward + bwd-data, halo-tiled "
                    "implicit GEMM with F(2,3) minimal filtering along the row: 12 of the direct form's 18 MFMA steps per pixel pair are executed"
                    + ("; layers below 128 source channels run conv3x3_halo_kernel, the direct form" if mode == "f16x2" else "") + f"; {how}; ")
        if not wino:
            what = f"conv3x3_halo_kernel<{tag}> (3x3 stride-1 forward + bwd-data, halo-tiled implicit GEMM; {how}; "
        out["roofline"] = entry(
            summ["conv3x3"],
            what + "achieved = ALGORITHMIC FLOP/s of the direct convolution (2*B*H*W*Cout*9*Cin per launch) over the C-ABI call incl. its weight "
            "pack pre-kernel, peak = dense 16-bit MFMA peak / MFMAs per product)", kname)
        # share of the direct form's products the matrix cores execute, FLOP-weighted over the launches of the pass: 1/2 on the F(4,3)
        # kernel, 2/3 on the F(2,3) kernels (from the per-launch records: entry point + leading integer arguments)
        frac_exec = None
        if wino and mode == "bf16x3":
            frac_exec = 2.0 / 3.0
            if timer is not None and timer.tags.get("conv3x3"):
                num = den = 0.0
                for (s_, e_, fl), t_ in zip(timer.records["conv3x3"], timer.tags["conv3x3"]):
                    bwd = t_[0].startswith("fs_conv2d_bwd_data")
                    W_, Cs_, Cd_ = t_[3], (t_[7] if bwd else t_[4]), (t_[4] if bwd else t_[7])
                    num += fl * (0.5 if (f43 and takes_f43(W_, Cs_, Cd_)) else 2.0 / 3.0); den += fl
                frac_exec = num / den if den else frac_exec
        out["roofline"]["executed_mfma_fraction"] = round(frac_exec, 4) if frac_exec is not None else None
        ka = _profiled_kernel_avg_us(kname) if mode == "bf16x3" else None
        out["roofline"]["kernel_avg_us"] = ka["kernel_avg_us"] if ka else None
        out["roofline"]["kernel_avg_us_source"] = (ka["source"] + " (rocprofv3 --kernel-trace --stats: the kernels alone; avg_launch_us is the in-process bracket "
                                                   "around the C-ABI call incl. the weight pack)") if ka else None
        out["roofline"]["note"] = ("frac prices the direct-convolution FLOPs against the dense peak; the matrix cores execute executed_mfma_fraction of them "
                                   "(frac * executed_mfma_fraction = share of the peak the MFMA pipe actually delivers)") if frac_exec is not None else None
        if "wgrad3x3" in summ:
            out["roofline_wgrad"] = entry(summ["wgrad3x3"], f"conv_wgrad_class_kernel<{tag},3,3> (3x3 stride-1 bwd-weight, 9 taps per workgroup)",
                                          f"conv_wgrad_class_kernel<fs_split::{tag}, 3, 3")
        if "conv_affine" in summ:
            out["roofline_other_convs"] = entry(
                summ["conv_affine"], f"conv_s2fwd_kernel<{tag}> / conv_s2bwd_kernel<{tag}> (3x3 stride-2 forward over the four input parity planes, bwd-data "
                                     f"with the four output parities in one launch; round 4) + conv1x1_gemm_kernel<{tag}> / conv_igemm_split_kernel<{tag}> "
                                     "(1x1 convs and the stride-4 head conv: HBM-bound at these channel counts)",
                [f"conv_s2fwd_kernel<fs_split::{tag}", f"conv_s2bwd_kernel<fs_split::{tag}", f"conv1x1_gemm_kernel<fs_split::{tag}", f"conv_igemm_split_kernel<fs_split::{tag}>"])
    elif "conv_affine" in summ:
        out["roofline"] = entry(summ["conv_affine"], "conv_igemm_affine_kernel<1> (fwd + bwd-data implicit GEMM, fp32 MFMA 32x32x2)", "conv_igemm_affine_kernel<1>")
        if "conv_wgrad" in summ:
            out["roofline_wgrad"] = entry(summ["conv_wgrad"], "conv_wgrad_taps_kernel<3|9> + conv_wgrad_kernel (bwd-weight, fp32 MFMA)", "conv_wgrad_taps_kernel")
    if "roofline" in out:
        dominant = (("conv3x3_wino4_kernel" if os.environ.get("FS_WINO4", "1") != "0" else "conv3x3_wino_kernel")
                    if (mode == "bf16x3" and os.environ.get("FS_WINOGRAD", "1") != "0") else "conv3x3_halo_kernel")
        out["roofline"]["mfma_utilisation"] = _profiled_mfma_busy(mode, dominant)       # offline SQ counters of the same kernel (rocprofv3 --pmc)
        out["roofline"]["serial_ms_per_step"] = round(1000.0 * serial_elapsed / nsteps, 2)
        out["roofline"]["measured"] = "second pass of the same K steps with branch streams serialised, HIP events per launch on the launch stream"
    if "bn_fwd" in summ:
        out["roofline_bn_fwd"] = hbm_entry(summ["bn_fwd"], "BatchNorm apply + residual + activation forward (algorithmic bytes = conv output read, [residual read,] "
                                           "activation written, 1 mask byte per 4 channels)", ["bn_act_fwd_kernel"])
    if "bn_bwd" in summ:
        out["roofline_bn_bwd"] = hbm_entry(summ["bn_bwd"], "BatchNorm + activation backward (partial sums where no producer formed them + finalize + apply; "
                                           "algorithmic bytes = dz, conv output and mask read once, dy [and the residual gradient] written once; "
                                           "traffic = bytes of the three kernels per layer)",
                                           ["bn_bwd_partial_kernel", "bn_bwd_finalize_kernel", "bn_bwd_apply_kernel"], per="layer")
    fe = {}
    for kind, (kname, desc) in FRONTEND_KINDS.items():
        if kind in summ:
            kk = summ[kind]
            gbs = kk["flops"] / (kk["total_ms"] * 1e-3) / 1e9
            us = 1000.0 * kk["total_ms"] / kk["launches"]
            fe[kind[3:]] = {"kernel": f"{kname} ({desc})", "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(gbs / PEAK_HBM_GBS, 4), "avg_launch_us": round(us, 2),
                            "mb_per_launch": round(kk["flops"] / kk["launches"] / 1e6, 3), "launches_per_step": kk["launches"] // nsteps}
            tr = _profiled_traffic(kname)
            fe[kind[3:]]["traffic"] = tr["hbm_bytes_per_launch"] if tr else None
            if tr is not None:            # what actually crossed the fabric (64-B sectors of a gather, re-reads): counter bytes / launch time
                fe[kind[3:]]["traffic_gbs"] = round(tr["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9, 1)
    if fe:
        fe["note"] = "algorithmic bytes (SURVEY.md 8(d)): every operand element the kernel needs read once, every result written once; the gather kernels touch 4 taps per output"
        out["frontend"] = fe
    return out


def family_entries(mode, timer, nsteps):
    """The strided / 1x1 convolution family by the bound the round-4 counters established: strided 3x3 layers against the
    mode's MFMA roof, 1x1 layers against the HBM roof with their algorithmic bytes (input read once, output written once, weights)."""
    peak = mode_peak(mode)
    groups = {"strided": [0, 0.0, 0.0], "strided_wgrad": [0, 0.0, 0.0], "pointwise": [0, 0.0, 0.0], "pointwise_wgrad": [0, 0.0, 0.0]}
    for kind in ("conv_affine", "conv_wgrad"):
        for (s, e, f), tag in zip(timer.records.get(kind, []), timer.tags.get(kind, [])):
            if tag is None or len(tag) < 13:
                continue
            B, H, W, Cin, Ho, Wo, Cout, R, S, stride = tag[1:11]
            ms = s.elapsed_time(e)
            wg = "_wgrad" if kind == "conv_wgrad" else ""
            if R == 1 and S == 1:
                d = groups["pointwise" + wg]
                d[2] += 4.0 * (B * H * W * Cin / (stride * stride) + B * Ho * Wo * Cout + Cin * Cout)
            elif stride > 1:
                d = groups["strided" + wg]
                d[2] += f
            else:
                continue
            d[0] += 1
            d[1] += ms
    out = {}
    for name, (n, ms, work) in groups.items():
        if n == 0 or ms <= 0:
            continue
        if name.startswith("pointwise"):
            gbs = work / (ms * 1e-3) / 1e9
            out["roofline_" + name] = {"kernel": "1x1 convolutions" + (" (bwd-weight: linear_wgrad_kernel)" if "wgrad" in name else " forward + bwd-data (conv1x1_gemm_kernel)") +
                                                 ": HBM-bound at 64-512 channels (profiles/r04/pmc); algorithmic bytes = input read once, output written once, weights",
                                       "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                                       "launches_per_step": n // nsteps, "avg_launch_us": round(1e3 * ms / n, 2), "ms_per_step": round(ms / nsteps, 3)}
        else:
            tf = work / (ms * 1e-3) / 1e12
            out["roofline_" + name] = {"kernel": "strided 3x3 convolutions" + (" bwd-weight (conv_wgrad_class / conv_wgrad_planes / gathered linear_wgrad kernels)" if "wgrad" in name
                                                 else " forward + bwd-data (conv_s2fwd_kernel, conv_s2bwd_kernel; the stride-4 head conv on conv1x1_gemm_kernel over gathered / scattered rows)"),
                                       "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                                       "launches_per_step": n // nsteps, "avg_launch_us": round(1e3 * ms / n, 2), "ms_per_step": round(ms / nsteps, 3)}
    return out


if __name__ == "__main__":
    main()
