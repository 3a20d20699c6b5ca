"""The ``train_arg`` region/argument branch of ``CLIP.forward`` (reference model_clip.py:430-488)."""
from __future__ import annotations

import torch

from .utils_image import patch_from_norm_bbox


def region_losses(model, grid_features, bboxs, bbox_desc_vec, bbox_label_vec, train_arg: str):
    raise NotImplementedError("region/argument branch: HIP bbox-pool kernel lands with ot.hip (next milestone)")
