"""clip_event_amd: MI355X-native drop-in for the CLIP-Event model.forward()/loss hot path.

Importing the package needs no GPU (tokenizer, label layout, synthetic data are host-side);
every compute entry point fails loudly when the HIP library or the GPU is missing."""
from .tokenizer import tokenize  # noqa: F401
from .utils_image import patch_from_norm_bbox  # noqa: F401

__all__ = ["tokenize", "patch_from_norm_bbox", "CLIP", "build_model", "CriterionContrastive", "CriterionAlignment"]


def __getattr__(name):
    if name in ("CLIP", "build_model"):
        from . import model
        return getattr(model, name)
    if name in ("CriterionContrastive", "CriterionAlignment"):
        from . import losses
        return getattr(losses, name)
    raise AttributeError(name)
