"""Plugin surface of the reference, backed by the HIP path.

`ModelBuilder` (models/models.py:1146-1230) and `DeformSegmentationModule` (models/models.py:476-1094)
with the same constructor/forward signatures, attribute names, return tuples, state_dict keys and the
`feed_dict['seg_label']` side effect (:951), under the effective LVIS-50 configuration
(SURVEY.md Appendix A).  Branches the default run never takes raise NotImplementedError.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import modules as M
from . import ops
from .weights import apply_name_keyed_init  # noqa: F401  (re-export for callers)


def make_gaussian_1d(size: int, fwhm: float) -> np.ndarray:
    """1-D factor of makeGaussian (models/models.py:140-157): G[i,j] = g[i]*g[j], float64."""
    ax = np.arange(0, size, 1, float)
    return np.exp(-4 * np.log(2) * (ax - size // 2) ** 2 / fwhm ** 2)


class ModelBuilder:
    @staticmethod
    def weights_init(m):
        """models/models.py:1149-1155 (by class name: Conv -> kaiming normal, BatchNorm -> w=1, b=1e-4)."""
        classname = m.__class__.__name__
        if classname.find("Conv2d") != -1 and hasattr(m, "weight"):
            nn.init.kaiming_normal_(m.weight.data)
        elif classname.find("BatchNorm") != -1:
            m.weight.data.fill_(1.0)
            m.bias.data.fill_(1e-4)

    @staticmethod
    def _load(net, weights):
        if len(weights) > 0:
            net.load_state_dict(torch.load(weights, map_location=lambda storage, loc: storage), strict=False)
        return net

    @staticmethod
    def build_encoder(arch="resnet50", fc_dim=2048, weights="", dilate_rate=4):
        arch = arch.lower()
        if arch == "hrnetv2_nodownsp":
            net = M.hrnetv2_nodownsp(pretrained=False)
        elif arch == "deeplab":
            from . import deeplab as _dl
            net = _dl.deeplab(pretrained=False)
        elif arch == "segformer":
            from . import segformer as _sf
            net = _sf.segformer(pretrained=False)
        else:
            raise Exception("Architecture undefined!")
        return ModelBuilder._load(net, weights)

    @staticmethod
    def build_decoder(arch="upernet", fc_dim=2048, num_class=150, weights="", use_softmax=False):
        arch = arch.lower()
        if arch == "c1":
            net = M.C1(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax)
        else:
            raise Exception("Architecture undefined!")
        net.apply(ModelBuilder.weights_init)
        return ModelBuilder._load(net, weights)

    @staticmethod
    def build_net_saliency(cfg=None, weights=""):
        if not (cfg.MODEL.track_running_stats and cfg.MODEL.saliency_net == "fovsimple"):
            raise Exception("Architecture undefined!")
        net = M.fov_simple(cfg)
        if len(weights) == 0:
            net.apply(ModelBuilder.weights_init)
        return ModelBuilder._load(net, weights)

    @staticmethod
    def build_net_compress(cfg=None, weights=""):
        net = M.CompressNet(cfg)
        if len(weights) == 0:
            net.apply(ModelBuilder.weights_init)
        return ModelBuilder._load(net, weights)


class _FilterHolder(nn.Module):
    """Keeps the `filter.weight` state_dict key of the reference (a fixed 91x91 Gaussian, Q6)."""

    def __init__(self, k, fwhm):
        super().__init__()
        g = make_gaussian_1d(k, fwhm)
        self.register_buffer("weight", torch.from_numpy(np.outer(g, g)).float().view(1, 1, k, k))


class DeformSegmentationModule(nn.Module):
    def __init__(self, net_encoder, net_decoder, net_saliency, net_compress, crit, cfg, deep_sup_scale=None):
        super().__init__()
        self.encoder = net_encoder
        self.decoder = net_decoder
        self.localization = net_saliency
        self.net_compress = net_compress
        self.cfg = cfg
        self.deep_sup_scale = deep_sup_scale
        T = cfg.TRAIN
        if cfg.MODEL.saliency_output_size_short != 0 or cfg.MODEL.gaussian_ap != 0.0:
            raise NotImplementedError("only saliency_output_size_short=0, gaussian_ap=0.0 (defaults) are built")
        self.grid_size_x, self.grid_size_y = int(T.saliency_input_size[0]), int(T.saliency_input_size[1])
        self.padding_size_x = self.padding_size_y = int(cfg.MODEL.gaussian_radius)
        self.input_size = tuple(T.saliency_input_size)
        self.input_size_net = tuple(T.task_input_size)
        ti, si = tuple(int(v) for v in self.input_size_net), tuple(int(v) for v in self.input_size)
        if ti != si and (ti[0] % si[0] or ti[1] % si[1]):
            raise NotImplementedError("task_input_size must be an integer multiple of saliency_input_size (grid up-sampling, models/models.py:621-631)")
        if cfg.DATASET.segm_downsampling_rate != 1:
            raise NotImplementedError("DATASET.segm_downsampling_rate must be 1 (grid_y == grid, models/models.py:627): with any other rate the "
                                      "reference samples the label at task_input_size // rate and multiplies it with cls_label.repeat(1, HS, WS) at the "
                                      "saliency size (:968) against a prediction at the task size -- a shape error with every encoder of this path")
        k = 2 * self.padding_size_x + 1
        self.filter = _FilterHolder(k, cfg.MODEL.gaussian_radius)
        self.register_buffer("g1d", torch.from_numpy(make_gaussian_1d(k, cfg.MODEL.gaussian_radius)), persistent=False)
        self._check_cfg()

    def _check_cfg(self):
        c = self.cfg
        off = [("MODEL.loss_at_high_res", c.MODEL.loss_at_high_res),
               ("MODEL.gt_gradient", c.MODEL.gt_gradient), ("TRAIN.opt_deform_LabelEdge", c.TRAIN.opt_deform_LabelEdge)]
        for name, val in off:
            if val:
                raise NotImplementedError(f"{name}=True is outside the built hot path (SURVEY.md Appendix A)")
        if c.MODEL.upsample and c.MODEL.rev_deform_interp != "nearest":
            raise NotImplementedError("MODEL.upsample needs rev_deform_interp='nearest' (the reference's 'tri' default calls an undefined name)")
        if c.MODEL.uniform_sample == "BI":
            raise NotImplementedError("MODEL.uniform_sample='BI' is unreachable in the reference with this fork's (B,1,H,W) labels: "
                                      "nn.Upsample(mode='bilinear') of y.float().unsqueeze(1) is handed a 5-D tensor (models/models.py:877); "
                                      "'' (learned sampling) and any other value (uniform saliency, config/defaults.py:69) are built")
        if c.TRAIN.def_saliency_pad_mode not in ops.PAD_MODES:
            # models/models.py:819-825 has no else branch: xs_hm stays unbound and line 845 raises the NameError subclass below
            raise UnboundLocalError("local variable 'xs_hm' referenced before assignment (TRAIN.def_saliency_pad_mode must be "
                                    "'replication', 'reflect' or 'zero', models/models.py:819-825)")
        if c.TRAIN.def_saliency_pad_mode == "reflect" and self.padding_size_x > min(self.grid_size_x, self.grid_size_y) - 1:
            raise NotImplementedError("def_saliency_pad_mode='reflect' needs gaussian_radius <= saliency side - 1 (F.pad refuses it too)")
        if not c.TRAIN.opt_deform_LabelEdge_norm:
            raise NotImplementedError("only the min/max-normalised edge loss is built")
        if self.deep_sup_scale is not None:
            raise NotImplementedError("deep supervision is not used on this path")

    # ---- stages (exposed for stage-wise parity tests) -------------------------------------------
    def saliency(self, x, focus):
        """x (B,3,H,W), focus (B,2) -> xs (B,1,hs,ws), x_low (B,hs,ws,5) NHWC."""
        x_low = ops.gaze_lowres(x, focus, self.grid_size_x, self.grid_size_y)
        s = self.localization.forward_nhwc(x_low)
        return self.net_compress.softmax_nhwc(s), x_low

    def create_grid(self, xs):
        """xs (B,1,hs,ws) -> grid (B,ht,wt,2) at the task network's input size; the padding of TRAIN.def_saliency_pad_mode
        ('replication' nn.ReplicationPad2d, 'reflect' / 'zero' F.pad) folded in (models/models.py:594-637,819-825).  When task_input_size != saliency_input_size the (hs,ws) grid is bilinearly up-sampled
        (nn.Upsample(size=input_size_net, mode='bilinear'), :621-631); grid_y is the same tensor (segm_downsampling_rate 1)."""
        grid = ops.GaussGrid.apply(xs, self.g1d, self.padding_size_x, ops.PAD_MODES[self.cfg.TRAIN.def_saliency_pad_mode])
        ht, wt = int(self.input_size_net[0]), int(self.input_size_net[1])
        if (ht, wt) != (grid.shape[1], grid.shape[2]):
            grid = ops.GridUpsample.apply(grid, ht, wt)
        return grid

    @torch.no_grad()
    def unwarp(self, pred, grid, seg_size):
        """Full-resolution prediction from the foveated one: the inverse grid of `create_grid(..., segSize, x_inv)`
        (models/models.py:639-655), `F.grid_sample(pred, grid_inv)` (:933) and the nearest-neighbour hole filling of
        `fillMissingValues_tensor(..., interp_mode='nearest')` (:159-286), all on the device (the
        'tri' default of the reference calls an undefined name).  pred (B,C,h,w) logical NCHW, grid (B,h,w,2) as returned by
        create_grid; returns (pred_full (B,C,H,W), hole mask (B,H,W))."""
        return ops.unwarp_nearest(pred.contiguous(), grid, int(seg_size[0]), int(seg_size[1]))

    # models/models.py:721 asserts `not torch.isnan(xs).any()` in the middle of the forward: a device->host read that drains the
    # stream once per step (2-3 % of the training step on MI355X, profiles/r02/nan_check_ab.txt).  The same flag is taken on the
    # device, copied to pinned host memory without blocking, and the AssertionError (same message) is raised when the flag is
    # next looked at: at the end of train.train_step / train.eval_step, at the next forward, or by check_nan().
    # FS_NAN_CHECK=sync restores the reference's in-place assert, FS_NAN_CHECK=0 drops the check.
    _nan_mode = os.environ.get("FS_NAN_CHECK", "defer")

    def _note_nan(self, xs):
        if self._nan_mode == "0":
            return
        if self._nan_mode == "sync" or not xs.is_cuda:
            assert not torch.isnan(xs).any(), "xs contains NaN values!"
            return
        st = self.__dict__.setdefault("_nan_state", {"host": torch.zeros(1, dtype=torch.bool).pin_memory(), "event": None})
        st["host"].copy_(torch.isnan(xs).any().reshape(1), non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record()

    def check_nan(self):
        """Raise the pending `xs contains NaN values!` assertion of the last forward, if any (see _note_nan)."""
        st = self.__dict__.get("_nan_state")
        if st is None or st["event"] is None:
            return
        st["event"].synchronize()
        st["event"] = None
        assert not bool(st["host"][0]), "xs contains NaN values!"

    def forward(self, feed_dict, *, writer=None, segSize=None, F_Xlr_acc_map=False, count=None, epoch=None,
                feed_dict_info=None, feed_batch_count=None, cur_iter=None, is_inference=False, rank=None):
        self.check_nan()
        ops.reset_step_state()
        ops.DDP_ACTIVE = ops.under_torch_ddp(self)      # wrapped by torch DDP: weight gradients go through AccumulateGrad (ops.py)
        if segSize is not None:
            raise NotImplementedError("forward(segSize=...): the reference has no inference branch left either -- its forward body is one "
                                      "`if segSize is None:` block (models/models.py:828-1094) and falls off the end, returning None, which its only "
                                      "caller (eval.py:178-180) cannot unpack; use is_inference=True (train.eval_step) for evaluation")
        cfg = self.cfg
        x = feed_dict["img_data"].contiguous()
        y = feed_dict["seg_label"]
        if y.dim() == 3:
            y = y.unsqueeze(1)
        y = y.float().contiguous()
        focus = feed_dict["focus_point"].float().contiguous()
        hs, ws = self.grid_size_x, self.grid_size_y

        xs, _ = self.saliency(x, focus)
        self._note_nan(xs)
        xs = ops.grad_probe(xs, "dxs_sum")                              # (ops.GRAD_TRACE: no-ops unless a test switched the recorder on)
        xs_grid = xs
        if cfg.MODEL.uniform_sample != "":
            # models/models.py:816-818 ('Saliency', config/defaults.py:69): the sampler runs on a uniform map; the edge loss keeps the learned
            # one (xs_our is cloned at :726, before this line) and the grid path hands the saliency net a zero gradient (d(xs*0)/dxs)
            xs_grid = xs * 0 + 1.0 / (self.grid_size_x * self.grid_size_y)
        grid = ops.grad_probe(self.create_grid(ops.grad_probe(xs_grid, "dxs_grid", alias=True)), "dgrid")

        joint = cfg.TRAIN.deform_joint_loss
        if joint:
            target = ops.area_pool(y, hs, ws)
            edge_loss = ops.EdgeLoss.apply(ops.grad_probe(xs, "dxs_edge", alias=True), target, 0.05 * float(cfg.TRAIN.edge_loss_scale))

        label = ops.grid_sample_label(y, grid.detach())
        x_sampled = ops.grad_probe(ops.GridSample.apply(x, grid), "dx_sampled")      # (B,hs,ws,3) NHWC
        feat = self.encoder.forward_nhwc(x_sampled)
        pred = self.decoder.forward_nhwc(feat)                         # (B,K,hs,ws)
        feed_dict["seg_label"] = label                                  # models/models.py:951

        cls = feed_dict["cls_label"].to(label.dtype)
        gt = label * cls[:, :, None] + (1 - label) * (cfg.DATASET.num_class - 1)
        out = ops.SegLoss.apply(pred, gt.contiguous(), 5.0)
        loss = out[0]
        if joint:
            loss = loss + edge_loss
        acc = out[3].detach()
        accs = (out[4].detach(), out[5].detach(), out[6].detach())
        if cfg.MODEL.upsample:
            # models/models.py:869-873,933-940,1074-1083: the loss stays at the sampled resolution, the four accuracies are taken at
            # FULL resolution on the prediction warped back through the inverse grid (never-claimed pixels filled from their
            # nearest claimed neighbour) against the original label map.  No gradient flows through this branch.
            with torch.no_grad():
                Hf, Wf = int(y.shape[2]), int(y.shape[3])
                pred_full, _hole = ops.unwarp_nearest(pred.detach().contiguous(), grid.detach(), Hf, Wf)
                y_hs = y[:, 0].long()
                gt_hs = y_hs * cls[:, :, None] + (1 - y_hs) * (cfg.DATASET.num_class - 1)
                full = ops.SegLoss.apply(pred_full, gt_hs.contiguous(), 5.0)
                acc, accs = full[3], (full[4], full[5], full[6])
        if joint:
            if not is_inference:
                return loss, acc, edge_loss
            return loss, acc, edge_loss, accs[0], accs[1], accs[2]
        if not is_inference:
            return loss, acc
        return loss, acc, accs[0], accs[1], accs[2]
