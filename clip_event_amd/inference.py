"""Zero-shot scoring on the HIP forward path: what the reference's salient-event selection does
with the model (src/preprocess/preprocess_description_contrastive.py:127-132): ``model(image, text)`` under
``no_grad``, softmax of ``logits_per_image`` over the candidate texts, best candidate and its probability."""
from __future__ import annotations

from typing import Tuple

import torch


@torch.no_grad()
def zero_shot(model, image: torch.Tensor, text: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``image`` [B,3,R,R], ``text`` [N,77] candidate descriptions shared by all images ->
    ``(scores [B], pred_idx [B], probs [B,N])``.  Uses the over-batch logits whatever the training setting."""
    saved = model.constrastive_overbatch
    model.constrastive_overbatch = True
    try:
        logits_per_image, _ = model(image, text)
    finally:
        model.constrastive_overbatch = saved
    probs = logits_per_image.softmax(dim=-1)
    scores, pred_idx = torch.max(probs, dim=-1)
    return scores, pred_idx, probs
