"""Pins taken from the reference itself (VERDICT r1 "next" #8), written as small JSON fixtures:

  g15_state_dict.json  -- key -> [shape, dtype] of the reference DeformSegmentationModule's state_dict (HRNetV2 + C1 +
                          fov_simple + CompressNet under the LVIS-50 configuration), in the reference's own order;
  g15_config.json      -- the EFFECTIVE configuration of the README.md:79 LVIS-50 training command: the reference's
                          config/defaults.py (imported with a yacs stand-in), merged with config/deform.yaml, merged with the
                          command-line overlay of README.md:79, exactly the sequence of train_deform_semantic.py:616-624.

Run once in the build container:  python tests/golden/make_pins.py   (imports /root/reference read-only; data only is stored)
"""
import ast
import importlib
import json
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402

# README.md:79 (the LVIS-50 train command), as KEY VALUE pairs
README_OVERLAY = ["TRAIN.task_input_size", "(80,80)", "DIR", "./ckpt/lvis_50cls_hr_net_train", "TRAIN.deform_joint_loss", "True",
                  "VAL.no_upsample", "True", "TRAIN.num_epoch", "150", "TRAIN.eval_per_epoch", "10", "TRAIN.checkpoint_per_epoch", "20",
                  "TRAIN.skip_train_for_eval", "False", "VAL.no_upsample", "True", "DATASET.dataset_marker_train", "sp60000",
                  "DATASET.dataset_marker_valid", "sp12000", "MODEL.gaussian_radius", "45", "TRAIN.saliency_input_size", "(80, 80)"]


def _decode(v):
    """yacs `_decode_cfg_value`: strings that are Python literals become the literal ('2e-5' -> 2e-05, '(80,80)' -> tuple)."""
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _merge(node, other):
    for k, v in other.items():
        if isinstance(v, dict) and isinstance(node.get(k), dict):
            _merge(node[k], v)
        else:
            v = _decode(v)
            if isinstance(node.get(k), tuple) and isinstance(v, list):
                v = tuple(v)
            node[k] = v


def _plain(node):
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (tuple, list)):
        return [_plain(v) for v in node]
    return node


def effective_config():
    MM = rh.load_reference()          # installs the yacs stand-in (attribute dict) among others
    del MM
    defaults = importlib.import_module("config.defaults")          # /root/reference/config/defaults.py
    cfg = defaults._C
    with open(os.path.join(rh.REF, "config", "deform.yaml")) as f:
        _merge(cfg, yaml.safe_load(f))                             # cfg.merge_from_file
    for key, val in zip(README_OVERLAY[0::2], README_OVERLAY[1::2]):       # cfg.merge_from_list
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = _decode(val)
    return cfg


def main():
    cfg = effective_config()
    with open(os.path.join(HERE, "g15_config.json"), "w") as f:
        json.dump(_plain(cfg), f, indent=1, sort_keys=True)
    print("wrote g15_config.json:", sum(len(v) if isinstance(v, dict) else 1 for v in cfg.values()), "values")

    import make_goldens as MG        # noqa: F401  (re-uses build_reference_module; its import loads the reference once more)
    m = MG.build_reference_module(rh.reference_cfg())
    sd = m.state_dict()
    table = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    with open(os.path.join(HERE, "g15_state_dict.json"), "w") as f:
        json.dump(table, f, separators=(",", ":"))
    print("wrote g15_state_dict.json:", len(table), "entries,", sum(int(__import__("numpy").prod(s)) for _, s, _ in table), "elements")


if __name__ == "__main__":
    main()
