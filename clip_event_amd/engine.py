"""The per-iteration body of the reference's ``train_one_epoch`` (engine.py:48-95) on the HIP path:
forward, contrastive criterion (+ optional OT alignment), loss SUM (engine.py:67), backward,
global-norm clip to 1 and Adam (fused, optim.py).  With more than one rank the features are
all-gathered for a global-batch InfoNCE and parameter gradients are averaged (distributed.py)."""
from __future__ import annotations

import logging
import math
import sys
from typing import Dict, Optional

import torch

from . import distributed as D
from .functional import logits_from_features
from .losses import CriterionAlignment, CriterionContrastive


def contrastive_step_losses(model, criterion: CriterionContrastive, image, text, labels_per_image, labels_per_text,
                            index_pos, global_batch: bool = True) -> Dict[str, torch.Tensor]:
    """Forward + criterion.  ``global_batch`` (W > 1): logits_per_image = s * I_local @ T_all^T,
    logits_per_text = s * T_local @ I_all^T with labels from ``distributed.global_labels``."""
    if D.active() and global_batch:
        fi, ft = model.encode_both(image, text)
        fi_all, ft_all = D.gather_feature_pair(fi, ft)
        overbatch = model.constrastive_overbatch
        lpi, _ = logits_from_features(fi, ft_all if overbatch else ft, model.logit_scale, overbatch, want="image")
        _, lpt = logits_from_features(fi_all, ft, model.logit_scale, True, want="text")
    else:
        lpi, lpt = model(image, text)
    return criterion(lpi, lpt, labels_per_image, labels_per_text, index_pos=index_pos,
                     constrastive_overbatch=model.constrastive_overbatch)


def train_step(model, criterion, optimizer, image, text, labels_per_image, labels_per_text, index_pos,
               grad_sync: Optional[D.GradSync] = None, criterion_ot: Optional[CriterionAlignment] = None,
               object_vec=None, entitytxt_vec=None, object_num=None, entitytxt_num=None,
               check_finite: bool = False) -> Dict[str, torch.Tensor]:
    """``check_finite`` reproduces engine.py:70-81 (all-rank mean of the losses, ``.item()``, stop on a non-finite
    value); it costs the host synchronisation the reference pays every step, so it is off by default."""
    optimizer.zero_grad()
    loss_dict = contrastive_step_losses(model, criterion, image, text, labels_per_image, labels_per_text, index_pos)
    passes = 1
    if model.alignment and criterion_ot is not None:
        image_features, text_features = model.sim_entity(object_vec, entitytxt_vec)       # engine.py:57-63
        loss_dict.update(criterion_ot(text_features, image_features, entitytxt_num, object_num))
        passes = 2
    losses = sum(loss for loss in loss_dict.values())                                      # engine.py:67
    if check_finite:
        reduced = D.reduce_dict({k: v.detach() for k, v in loss_dict.items()})
        loss_value = float(sum(v for v in reduced.values()))
        if not math.isfinite(loss_value):
            logging.error("Loss is {}, stopping training".format(loss_value))
            logging.error(reduced)
            sys.exit(1)
    losses.backward()
    if grad_sync is not None:
        grad_sync.finish(passes_per_tower=passes)
    optimizer.step()                                                                       # clip + Adam
    return loss_dict
