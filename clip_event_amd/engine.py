"""The per-iteration body of the reference's ``train_one_epoch`` (engine.py:48-95) on the HIP path:
forward, contrastive criterion (+ optional OT alignment), loss SUM (engine.py:67), backward,
global-norm clip to 1 and Adam (fused, optim.py).  With more than one rank the features are
all-gathered for a global-batch InfoNCE and parameter gradients are averaged (distributed.py)."""
from __future__ import annotations

import functools
import logging
import math
import operator
import os
import sys
from typing import Dict, Optional

import torch

from . import distributed as D
from .functional import attach_lengths, fused_contrastive_losses, fused_head_ok, logits_from_features, tokens_to_device
from .losses import CriterionAlignment, CriterionContrastive


def _unwrap(model):
    """``DistributedDataParallel(model)`` (ours, train.py:222-225) -> the CLIP module (engine.py:52,57-61)."""
    return model.module if isinstance(model, D.DistributedDataParallel) else model


def contrastive_step_losses(model, criterion: CriterionContrastive, image, text, labels_per_image, labels_per_text,
                            index_pos, global_batch: bool = True, train_arg=None, bboxs=None, bbox_desc_vec=None,
                            bbox_label_vec=None) -> Dict[str, torch.Tensor]:
    """Forward + criterion.  ``global_batch`` (W > 1): logits_per_image = s * I_local @ T_all^T,
    logits_per_text = s * T_local @ I_all^T with labels from ``distributed.global_labels``.  With ``train_arg``
    (model_clip.py:419-488) the two region / argument losses join the dict as ``loss_bbox`` / ``loss_arg``; they
    are per-image sums over the local batch, no exchange."""
    model = _unwrap(model)
    extra = {}
    # fused head (no logits matrix in HBM): the 'ce' criterion over the batch, SURVEY 8(a) a6.  CE_FUSED_HEAD=0 keeps the
    # logits + criterion path (the one the reference API exposes; same values).
    # The fused head exists to keep a big logits matrix out of HBM (config 3 per rank: 512 x 20,480); at B = 256, K = 1 the
    # matrix is 256 KB and the plain path's kernels are the faster ones (same-box A/B: 12.75 / 12.53 vs 12.79 / 12.67 ms/step), so
    # the fused head takes over from 2^17 logits per direction (256 x 1280 and 512 x 512: fused ahead by 0.1-0.2 ms).  CE_FUSED_HEAD=1 / 0 forces either.
    force = os.environ.get("CE_FUSED_HEAD", "")
    logits = int(image.shape[0]) * int(text.shape[0]) * (D.world_size() if D.active() and global_batch else 1)
    fused = (force != "0" and (force == "1" or logits >= (1 << 17)) and getattr(criterion, "kind", None) == "ce"
             and model.constrastive_overbatch and fused_head_ok(model.embed_dim))
    dist_global = D.active() and global_batch
    if dist_global or fused:
        if train_arg is None:
            fi, ft = model.encode_both(image, text)
        else:
            fi, ft, loss_bbox, loss_arg = model.encode_with_regions(image, text, train_arg, bboxs, bbox_desc_vec,
                                                                    bbox_label_vec)
            extra = {"loss_bbox": loss_bbox, "loss_arg": loss_arg}
        fi_all, ft_all = D.gather_feature_pair(fi, ft) if dist_global else (fi, ft)
        if fused:
            loss_dict = fused_contrastive_losses(fi, ft, fi_all, ft_all, model.logit_scale, labels_per_image,
                                                 labels_per_text, index_pos)
            loss_dict.update(extra)
            return loss_dict
        overbatch = model.constrastive_overbatch
        lpi, _ = logits_from_features(fi, ft_all if overbatch else ft, model.logit_scale, overbatch, want="image")
        _, lpt = logits_from_features(fi_all, ft, model.logit_scale, True, want="text")
    elif train_arg is None:
        lpi, lpt = model(image, text)
    else:
        lpi, lpt, loss_bbox, loss_arg = model(image, text, train_arg, bboxs, bbox_desc_vec, bbox_label_vec)
        extra = {"loss_bbox": loss_bbox, "loss_arg": loss_arg}
    loss_dict = criterion(lpi, lpt, labels_per_image, labels_per_text, index_pos=index_pos,
                          constrastive_overbatch=model.constrastive_overbatch)
    loss_dict.update(extra)
    return loss_dict


def train_step(model, criterion, optimizer, image, text, labels_per_image, labels_per_text, index_pos,
               grad_sync: Optional[D.GradSync] = None, criterion_ot: Optional[CriterionAlignment] = None,
               object_vec=None, entitytxt_vec=None, object_num=None, entitytxt_num=None,
               check_finite: bool = False, train_arg=None, bboxs=None, bbox_desc_vec=None,
               bbox_label_vec=None, text_lengths=None) -> Dict[str, torch.Tensor]:
    """One iteration of engine.py:48-95.  BASELINE config 4 is this call with ``criterion_ot`` + the object / entity
    tensors (``model.alignment``) and ``train_arg`` + boxes in ONE step, as the reference's forward takes them
    (model_clip.py:419-528, engine.py:57-63).  ``check_finite`` reproduces engine.py:70-81 (all-rank mean of the
    losses, ``.item()``, stop on a non-finite value); it costs the host synchronisation the reference pays every
    step, so it is off by default.

    Token tensors may arrive on the HOST, as the reference's data loader yields them (engine.py:52 copies them): their
    caption lengths are then taken on the host and travel with the asynchronous copy, so the text tower never reads
    anything back from the device.  For tokens already on the GPU pass ``text_lengths`` (host integers, tokens up to and
    including the EOT) or tag the tensor with ``functional.attach_lengths``; without either the lengths are read back
    (one small synchronous copy per new tensor)."""
    wrapped = model
    model = _unwrap(model)
    dev = next(model.parameters()).device
    if text_lengths is not None:
        attach_lengths(text, text_lengths)
    text = tokens_to_device(text, dev)
    if entitytxt_vec is not None:
        entitytxt_vec = tokens_to_device(entitytxt_vec, dev)
    if grad_sync is None and wrapped is not model:
        grad_sync = wrapped.grad_sync
    if hasattr(optimizer, "zero_grad_first_touch"):
        optimizer.zero_grad_first_touch()      # nothing reads .grad between here and backward(): block weights are overwritten
    else:
        optimizer.zero_grad()
    loss_dict = contrastive_step_losses(model, criterion, image, text, labels_per_image, labels_per_text, index_pos,
                                        train_arg=train_arg, bboxs=bboxs, bbox_desc_vec=bbox_desc_vec,
                                        bbox_label_vec=bbox_label_vec)
    if model.alignment and criterion_ot is not None:
        image_features, text_features = model.sim_entity(object_vec, entitytxt_vec)       # engine.py:57-63
        loss_dict.update(criterion_ot(text_features, image_features, entitytxt_num, object_num))
    losses = functools.reduce(operator.add, loss_dict.values())                            # engine.py:67 (sum() would add an int 0 first: one more launch)
    if check_finite:
        reduced = D.reduce_dict({k: v.detach() for k, v in loss_dict.items()})
        loss_value = float(sum(v for v in reduced.values()))
        if not math.isfinite(loss_value):
            logging.error("Loss is {}, stopping training".format(loss_value))
            logging.error(reduced)
            sys.exit(1)
    losses.backward()
    if check_finite and getattr(model, "stream16", False):
        # the fp16 streams clamp instead of overflowing, so a clipped activation / gradient never shows up as a non-finite
        # loss: read the device-side clamp counters on the synchronisation this mode pays anyway, before the update is applied
        sat_f, sat_g = model.stream16_saturation()
        if sat_f or sat_g:
            logging.error("fp16 stream saturated ({} forward, {} gradient slots), stopping training".format(sat_f, sat_g))
            sys.exit(1)
    if hasattr(model, "_settle_first_touch"):
        model._settle_first_touch()    # a tower without a backward pass in this step: its weight gradients are zero
    if grad_sync is not None:
        grad_sync.finish()             # no-op when the autograd final callback has already run it
    optimizer.step()                                                                       # clip + Adam
    return loss_dict
