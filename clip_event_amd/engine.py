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

import numpy as np
import torch

from . import distributed as D
from .functional import (attach_lengths, fused_contrastive_losses, fused_head_ok, logits_from_features, small_contrastive_losses,
                         small_head_ok, tokens_to_device)
from .losses import CriterionAlignment, CriterionContrastive


def _unwrap(model):
    """``DistributedDataParallel(model)`` (ours, train.py:222-225) -> the CLIP module (engine.py:52,57-61)."""
    return model.module if isinstance(model, D.DistributedDataParallel) else model


def _contrastive_from_features(model, criterion, fi, ft, labels_per_image, labels_per_text, index_pos, global_batch):
    """Loss dict of the contrastive criterion from the two towers' (raw) features: all-gather over the ranks when a
    process group is live, then the fused head (no logits matrix in HBM) or logits + criterion."""
    # fused head: the 'ce' criterion over the batch, SURVEY 8(a) a6.  It exists to keep a big logits matrix out of HBM (config
    # 3 per rank: 512 x 20,480); at B = 256, K = 1 the matrix is 256 KB and the plain path's kernels are the faster ones
    # (same-box A/B: 12.75 / 12.53 vs 12.79 / 12.67 ms/step), so the fused head takes over from 2^17 logits per direction (256 x
    # 1280 and 512 x 512: fused ahead by 0.1-0.2 ms).  CE_FUSED_HEAD=1 / 0 forces either.
    force = os.environ.get("CE_FUSED_HEAD", "")
    dist_global = D.active() and global_batch
    logits = int(fi.shape[0]) * int(ft.shape[0]) * (D.world_size() if dist_global else 1)
    fused = (force != "0" and (force == "1" or logits >= (1 << 17)) and getattr(criterion, "kind", None) == "ce"
             and model.constrastive_overbatch and fused_head_ok(model.embed_dim))
    # below that size, in one process: the three-launch head (CE_SMALL_HEAD=0 or CE_FUSED_HEAD=0: the general logits + criterion path)
    if (not dist_global and not fused and force == "" and os.environ.get("CE_SMALL_HEAD", "1") != "0"
            and getattr(criterion, "kind", None) == "ce" and model.constrastive_overbatch and index_pos is not None
            and small_head_ok(fi, ft, index_pos)):
        return small_contrastive_losses(fi, ft, model.logit_scale, labels_per_image, labels_per_text, index_pos)
    fi_all, ft_all = D.gather_feature_pair(fi, ft) if dist_global else (fi, ft)
    if fused:
        return fused_contrastive_losses(fi, ft, fi_all, ft_all, model.logit_scale, labels_per_image, labels_per_text, index_pos)
    overbatch = model.constrastive_overbatch
    if dist_global:
        lpi, _ = logits_from_features(fi, ft_all if overbatch else ft, model.logit_scale, overbatch, want="image")
        _, lpt = logits_from_features(fi_all, ft, model.logit_scale, True, want="text")
    else:
        lpi, lpt = logits_from_features(fi, ft, model.logit_scale, overbatch)
    return criterion(lpi, lpt, labels_per_image, labels_per_text, index_pos=index_pos, constrastive_overbatch=overbatch)


def contrastive_step_losses(model, criterion: CriterionContrastive, image, text, labels_per_image, labels_per_text,
                            index_pos, global_batch: bool = True, train_arg=None, bboxs=None, bbox_desc_vec=None,
                            bbox_label_vec=None) -> Dict[str, torch.Tensor]:
    """Forward + criterion.  ``global_batch`` (W > 1): logits_per_image = s * I_local @ T_all^T,
    logits_per_text = s * T_local @ I_all^T with labels from ``distributed.global_labels``.  With ``train_arg``
    (model_clip.py:419-488) the two region / argument losses join the dict as ``loss_bbox`` / ``loss_arg``; they
    are per-image sums over the local batch, no exchange."""
    model = _unwrap(model)
    extra = {}
    if train_arg is None:
        fi, ft = model.encode_both(image, text)
    else:
        fi, ft, loss_bbox, loss_arg = model.encode_with_regions(image, text, train_arg, bboxs, bbox_desc_vec, bbox_label_vec)
        extra = {"loss_bbox": loss_bbox, "loss_arg": loss_arg}
    loss_dict = _contrastive_from_features(model, criterion, fi, ft, labels_per_image, labels_per_text, index_pos, global_batch)
    loss_dict.update(extra)
    return loss_dict


# How far the host may run ahead of the GPU.  Nothing in a step needs the host to wait (with host-side caption lengths the text
# tower no longer reads anything back), but an unthrottled host queues the whole run and that measured SLOWER: 12.93-12.94
# ms/step against 12.48-12.56 with the per-step read-back (which happened to hold the host about a step behind), same box.
# CE_STEPS_AHEAD = n (default 2; 0 = unlimited) makes the distance explicit: before enqueueing a step the host waits for the
# end of the step n before it.  Same-box A/B: n = 1 12.76-13.0 (the GPU drains at every step boundary), n = 2 / 3 / 4 12.50-12.56
# (second box 12.28-12.34, all three equal).
_STEPS_AHEAD = int(os.environ.get("CE_STEPS_AHEAD", "2"))


def _throttle(model):
    if _STEPS_AHEAD <= 0 or not torch.cuda.is_available():
        return
    evs = model.__dict__.setdefault("_step_events", [])
    while len(evs) >= _STEPS_AHEAD:
        evs.pop(0).synchronize()


def _mark_step_end(model):
    if _STEPS_AHEAD <= 0 or not torch.cuda.is_available():
        return
    ev = torch.cuda.Event()
    ev.record()
    model.__dict__.setdefault("_step_events", []).append(ev)


def _cat_tokens(parts):
    """One [n, T] token matrix for one text-tower pass; the host-side lengths travel along when every part has them."""
    if len(parts) == 1:
        return parts[0]
    out = torch.cat(parts, dim=0)
    lens = [getattr(p, "_ce_lengths", None) for p in parts]
    if all(l is not None for l in lens):
        attach_lengths(out, np.concatenate([np.asarray(l).reshape(-1) for l in lens]))
    return out


def merged_step_losses(model, criterion, criterion_ot, image, text, labels_per_image, labels_per_text, index_pos,
                       train_arg=None, bboxs=None, bbox_desc_vec=None, bbox_label_vec=None, object_vec=None,
                       entitytxt_vec=None, object_num=None, entitytxt_num=None, global_batch: bool = True):
    """BASELINE config 4's forward with ONE pass per tower.  The reference (and ``CLIP.forward`` / ``sim_entity`` here) runs
    the image tower twice (batch, object crops: engine.py:57-63) and the text tower up to four times (captions, entity
    mentions, role descriptions, role labels: model_clip.py:430-455, :531-552).  Tower rows are independent of each other,
    so the same values come out of one image pass over [images | crops] and one text pass over [captions | mentions | roles |
    labels]: a 64-image pass fills a quarter of the GPU and a 100-caption pass is ~200 launches of a few microseconds each,
    while the merged passes fill their tiles, and each tower's weight gradients are written once instead of accumulated.
    With ``train_arg`` the image pass keeps every token (the region branch pools the grid of the batch images; the crops'
    CLS features are row 0 of theirs)."""
    from .region import region_losses_from_features, region_plan
    dev = image.device
    B = image.shape[0]
    want_align = criterion_ot is not None and model.alignment
    images, texts = [image], [text]
    n_obj = n_ent = 0
    if want_align:
        n_obj, n_ent = object_vec.size(1), entitytxt_vec.size(1)
        images.append(object_vec.reshape(B * n_obj, *object_vec.shape[2:]))
        ent = entitytxt_vec.reshape(B * n_ent, entitytxt_vec.size(2))
        if getattr(entitytxt_vec, "_ce_lengths", None) is not None:
            attach_lengths(ent, entitytxt_vec._ce_lengths)
        texts.append(ent)
    plan = region_plan(model, bboxs, bbox_desc_vec, bbox_label_vec, train_arg, dev) if train_arg is not None else None
    if plan is not None:
        texts.append(plan.descs)
        if plan.use_label:
            texts.append(plan.labs)
    img_all = torch.cat([_f for _f in images], dim=0) if len(images) > 1 else image
    txt_all = _cat_tokens(texts)
    fi_all, ft_all = model.encode_both(img_all, txt_all, use_grid=train_arg is not None)
    loss_dict = {}
    # split the features back (views: autograd sums their gradients into one tensor per tower)
    n_cap = text.shape[0]
    ft = ft_all[:n_cap]
    pos = n_cap
    if train_arg is not None:
        pn = model.visual.patch_num
        fi = fi_all[:B, 0, :]
        grid = fi_all[:B, 1:, :].reshape(B, pn, pn, -1)
        obj_f = fi_all[B:, 0, :]
    else:
        fi, grid, obj_f = fi_all[:B], None, fi_all[B:]
    loss_dict.update(_contrastive_from_features(model, criterion, fi, ft, labels_per_image, labels_per_text, index_pos, global_batch))
    ent_f = None
    if want_align:
        ent_f = ft_all[pos:pos + B * n_ent].view(B, n_ent, -1)
        pos += B * n_ent
    if train_arg is not None:
        if plan is None:
            zero = torch.zeros((), dtype=torch.float32, device=dev)
            loss_dict.update(loss_bbox=zero, loss_arg=zero)
        else:
            nd = plan.descs.shape[0]
            desc_f = ft_all[pos:pos + nd]
            lab_f = ft_all[pos + nd:pos + 2 * nd] if plan.use_label else None
            lb, la = region_losses_from_features(model, grid, plan, desc_f, lab_f)
            loss_dict.update(loss_bbox=lb, loss_arg=la)
    if want_align:
        loss_dict.update(criterion_ot(ent_f, obj_f.reshape(B, n_obj, -1), entitytxt_num, object_num))      # engine.py:57-63
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
    _throttle(model)
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
    want_align = model.alignment and criterion_ot is not None
    if (want_align or train_arg is not None) and os.environ.get("CE_MERGE_PASSES", "1") != "0":
        # config 4: one pass per tower over [images | object crops] and [captions | entity mentions | role texts]
        loss_dict = merged_step_losses(model, criterion, criterion_ot if want_align else None, image, text, labels_per_image,
                                       labels_per_text, index_pos, train_arg=train_arg, bboxs=bboxs, bbox_desc_vec=bbox_desc_vec,
                                       bbox_label_vec=bbox_label_vec, object_vec=object_vec, entitytxt_vec=entitytxt_vec,
                                       object_num=object_num, entitytxt_num=entitytxt_num)
    else:
        loss_dict = contrastive_step_losses(model, criterion, image, text, labels_per_image, labels_per_text, index_pos,
                                            train_arg=train_arg, bboxs=bboxs, bbox_desc_vec=bbox_desc_vec,
                                            bbox_label_vec=bbox_label_vec)
        if want_align:
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
    _mark_step_end(model)
    return loss_dict
