"""yacs-compatible configuration node for the FovealSeg hot path.

The reference drives everything from a global mutable yacs ``cfg`` (config/defaults.py:1-247 +
config/deform.yaml:1-62 + CLI overlay, train_deform_semantic.py:616-624).  yacs is not installed
here, so `CfgNode` is a small attribute dict with the same `merge_from_list` / attribute access, and
`lvis50_cfg()` returns the *effective* configuration of the README LVIS-50 command
(README.md:79) restricted to the keys the hot path reads.
"""
import ast
import copy


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_list(self, kv):
        assert len(kv) % 2 == 0, "expected KEY VALUE pairs"
        for key, val in zip(kv[0::2], kv[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = val
        return self


def lvis50_cfg() -> CfgNode:
    C = CfgNode()
    C.DIR = "ckpt/lvis_50cls_hr_net_train"
    C.DATASET = CfgNode(
        num_class=51, segm_downsampling_rate=1, grid_path="", binary_class=-1,
        list_train="", root_dataset="")
    C.MODEL = CfgNode(
        arch_encoder="hrnetv2_nodownsp", arch_decoder="c1", fc_dim=960,
        weights_encoder="", weights_decoder="", weights_net_saliency="", weights_net_compress="",
        saliency_net="fovsimple", track_running_stats=True,
        gaussian_radius=45, gaussian_ap=0.0, saliency_output_size_short=0,
        uniform_sample="", gt_gradient=False, fix_gt_gradient=False, ignore_gt_labels=[],
        gt_grad_gaussian_blur_r=1, gt_gradient_intrinsic_only=False,
        loss_at_high_res=False, upsample=False, rev_deform_interp="nearest")
    C.TRAIN = CfgNode(
        saliency_input_size=(80, 80), task_input_size=(80, 80), task_input_size_eval=(),
        dynamic_task_input=(1, 1), batch_size_per_gpu=64, num_gpus=1,
        deform_joint_loss=True, opt_deform_LabelEdge=False, opt_deform_LabelEdge_norm=True,
        opt_deform_LabelEdge_softmax=False, edge_loss_scale=100.0, fixed_edge_loss_scale=-1.0,
        stage_adjust_edge_loss=1.0, deform_zero_bound=True, deform_zero_bound_factor=1,
        deform_pretrain_bol=True, deform_pretrain=100,
        smooth_deform_2nd_start=2001, smooth_deform_2nd_end=2001,
        fix_seg_start_epoch=2000, fix_seg_end_epoch=2001,
        fix_deform_aft_pretrain=False, fix_deform_start_epoch=2000, fix_deform_end_epoch=2001,
        def_saliency_pad_mode="replication", global_epoch=1,
        optim="adam", lr_encoder=2e-5, lr_decoder=2e-5, lr_foveater=2e-5, lr_pow=0.9,
        lr_mult_encoder=0.001, lr_mult_decoder=0.001, lr_mult_saliency=0.001, lr_mult_compress=0.001,
        weight_decay=1e-4, beta1=0.9, fix_bn=False, scale_by_iter=False,
        fov_scale_lr="", fov_scale_pow=1, fov_scale_seg_only=False,
        epoch_iters=744, num_epoch=150, start_epoch=0, max_iters=744 * 150, disp_iter=20, seed=304,
        running_lr_encoder=2e-5, running_lr_decoder=2e-5, running_lr_foveater=2e-5)
    C.VAL = CfgNode(y_sampled_reverse=False, no_upsample=True, checkpoint="epoch_last.pth")
    return C
