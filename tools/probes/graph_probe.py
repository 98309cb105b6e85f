"""Capture one configs[4] training step (DeepLab, B = 16, 2048^2) in a HIP graph and replay it -- stage by stage, with faulthandler on, so
that a fault inside the capture names its stage.  Dropout off and learning rate / Adam step baked into the captured launches (timing and
capturability only).      python tools/probes/graph_probe.py config4 [forward|fwd_bwd|step]      GRAPH_SAME_STREAM=1: capture on the warm-up stream

Round 5 (ROCm 7.2 / torch 2.10.0+rocm7.0; profiles/r05/graph_probe.txt).  Round 4's "host segmentation fault inside the capture of the backward"
is a fault inside hipStreamEndCapture (torch.cuda.graphs.capture_end), with two independent causes:
 1. the outputs of the last EAGER step were still alive: they hold that step's autograd graph and with it the AccumulateGrad nodes of the
    parameters that go through autograd, created on the stream that step ran on -- the legacy default stream.  The capture reuses those nodes
    (torch warns about the stream mismatch), the engine orders the default stream against the capturing stream, and the default stream
    cannot join a capture.  Dropping the outputs first (below) makes forward + backward capture and replay;
 2. the side-stream weight gradients: the side stream is forked into the capture once per small layer (~300 times) and joined once at the
    end; hipStreamEndCapture faults on that (not an event re-recorded inside the capture: a ring large enough to avoid it changes nothing).
    ops.py keeps those launches on the node's stream while a capture is on.
What the capture is worth: forward + backward replay in 27.4 ms against 28.4 ms for the whole eager step -- with the host out of the way the
dependent chain of ~1 700 small launches is the step.  The host's 27 ms to enqueue (round 4) and the GPU's 27 ms to execute are the same
size; a graph only pays once the small layers overlap INSIDE it (graph-level forks hipStreamEndCapture accepts).  Correct replays would also
need the per-step scalars that are kernel arguments today (dropout key, learning rate, Adam's step count) to come from device words."""
import faulthandler, os, sys, time
faulthandler.enable(all_threads=True)          # a host fault inside the capture prints every thread's Python stack
os.environ["FS_NAN_CHECK"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fovealseg
from fovealseg import ops, train as T
which = sys.argv[1] if len(sys.argv) > 1 else "config4"
cfg = fovealseg.lvis50_cfg()
B, size = 16, 2048
if which == "config4":
    cfg.MODEL.arch_encoder = "deeplab"
elif which == "headline":
    B, size = 64, 1024
dev = torch.device("cuda", 0)
module, nets = T.build_module(cfg, device=dev)
module.train()
for m in module.modules():
    if hasattr(m, "drop_p"):
        m.drop_p = 0.0
    if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
        m.p = 0.0
opts = T.create_optimizers(nets, cfg)
batch = T.synthetic_batch(B, size, size, seed=1, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=i)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5):
    out = T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=3 + i)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 5
print(f"eager {1e3 * eager:.2f} ms/step  loss {float(out[0].detach()):.5f}", flush=True)
# The outputs of the last eager step hold its autograd graph, and with it the AccumulateGrad nodes of the parameters that go through
# autograd -- created on the stream that step ran on (the legacy default stream).  A capture would reuse those nodes, the engine would
# order that stream against the capturing one, and the default stream cannot join a capture: drop the graph first.
del out
stage = sys.argv[2] if len(sys.argv) > 2 else "step"
X, Fp, Y, cls = batch
feed = {"img_data": X[:, :3], "seg_label": Y, "focus_point": Fp, "cls_label": cls}
g = torch.cuda.CUDAGraph()
same = os.environ.get("GRAPH_SAME_STREAM", "0") == "1"      # capture on the warm-up stream: AccumulateGrad nodes and capture share one stream
print(f"capturing: {stage}  (capture on the warm-up stream: {same}, side-stream weight gradients below {ops.WGRAD_SIDE_FLOPS / 1e9:g} GFLOP)", flush=True)
with torch.cuda.graph(g, stream=s if same else None):
    if stage == "forward":                  # bisecting a capture fault: forward only / forward + backward / the whole step
        with torch.no_grad():
            out = module(dict(feed), epoch=1, cur_iter=8)
    elif stage == "fwd_bwd":
        for o in opts:
            o.zero_grad()
        out = module(dict(feed), epoch=1, cur_iter=8)
        out[0].mean().backward()
    else:
        out = T.train_step(module, opts, batch, cfg, epoch=1, cur_iter=8)
torch.cuda.synchronize()
print("captured", flush=True)
for i in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    g.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / 10
print(f"replay {1e3 * rep:.2f} ms/step  loss {float(out[0]):.5f}  speed-up {eager / rep:.2f}x  -> {B / rep:.1f} img/s", flush=True)
