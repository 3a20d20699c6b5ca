"""Integer bbox -> patch-grid helper of the region/argument branch (reference utils_image.py:28-32)."""
import math


def patch_from_norm_bbox(bbox_norm, patch_size: int = 7):
    """Normalised (x_min, y_min, x_max, y_max) in [0,1] -> grid indices: floor of the minimum
    corner, ceil of the maximum corner, each scaled by ``patch_size``.  Bit-exact integer contract."""
    x_min, y_min, x_max, y_max = bbox_norm
    return (math.floor(x_min * patch_size), math.floor(y_min * patch_size),
            math.ceil(x_max * patch_size), math.ceil(y_max * patch_size))
