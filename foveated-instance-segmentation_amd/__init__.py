"""fovealseg -- MI355X-native FovealSeg forward/backward hot path.

Host side mirrors the reference plugin surface (`ModelBuilder`, `DeformSegmentationModule`;
/root/reference/models/models.py:476-1230); every op below that surface is a hand-written HIP
kernel for gfx950 reached through the C-ABI library declared in include/fovealseg.h.
"""
from . import config, weights, hip, ops, modules, models, data  # noqa: F401
from .models import ModelBuilder, DeformSegmentationModule  # noqa: F401
from .config import lvis50_cfg  # noqa: F401

__all__ = ["config", "weights", "hip", "ops", "modules", "models", "data", "ModelBuilder", "DeformSegmentationModule", "lvis50_cfg"]
