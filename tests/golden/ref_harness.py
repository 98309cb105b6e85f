"""Import the read-only reference (/root/reference) on CPU with stubs for the packages this image
lacks.  Used ONLY by tests/golden/make_goldens.py in the build container; nothing on the GPU box
imports it (the reference does not travel).  Recipe: SURVEY.md §8(c).
"""
import importlib.machinery
import os
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def load_reference():
    """Returns the reference `models.models` module, importable on CPU."""
    import torch
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    # transformers must be imported before a torchvision stub exists
    try:
        import transformers  # noqa: F401
        from transformers import SegformerForSemanticSegmentation, SegformerConfig  # noqa: F401
    except Exception:
        pass
    repo_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(repo_root, "oracle"))
    from fovealseg_oracle import dice_loss_multiclass

    class DiceLoss(torch.nn.Module):                     # toolbelt restatement (see oracle header)
        def __init__(self, mode, *a, **k):
            super().__init__()
            assert mode == "multiclass"

        def forward(self, y_pred, y_true):
            return dice_loss_multiclass(y_pred, y_true)

    tv = _stub("torchvision")
    _stub("torchvision.utils")
    tvm = _stub("torchvision.models")
    seg = _stub("torchvision.models.segmentation", DeepLabV3_ResNet50_Weights=object, deeplabv3_resnet101=None)
    tvm.segmentation = seg
    tv.models = tvm
    _stub("torchvision.transforms")
    tv.utils = sys.modules["torchvision.utils"]
    _stub("torchsnooper")
    _stub("cv2")
    _stub("albumentations")
    _stub("segmentation_models_pytorch")
    _stub("peft", get_peft_model=None, LoraConfig=None, TaskType=None)
    _stub("pytorch_toolbelt")
    _stub("pytorch_toolbelt.losses")
    _stub("pytorch_toolbelt.losses.dice", DiceLoss=DiceLoss)
    _stub("yacs")
    _stub("yacs.config", CfgNode=_CfgNode)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.reset_max_memory_allocated = lambda *a, **k: None

    import models.models as MM
    _orig = MM.gen_grid_mtx_2xHxW
    MM.gen_grid_mtx_2xHxW = lambda H, W, device=None: _orig(H, W, device=None)
    return MM


def reference_cfg():
    """deform.yaml + README.md:79 overrides over config/defaults.py, as an attribute dict."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import fovealseg  # noqa: F401
    from fovealseg.config import lvis50_cfg
    c = lvis50_cfg()
    c.TRAIN.global_epoch = 1
    c.DATASET.grid_path = ""
    return c
