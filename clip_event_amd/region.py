"""The ``train_arg`` region/argument branch of ``CLIP.forward`` (reference model_clip.py:430-488).

The reference loops over images, slices the grid features per bounding box, mean-pools them and
runs ``encode_text`` once per image; it then calls ``self.loss_func``, which it never defines
 -- here ``loss_func`` is cross-entropy.  This implementation pools every box of the
batch in one launch and encodes all role descriptions in ONE text-tower pass (same values: the
tower is per-row), then forms the small per-image InfoNCE terms.  Quirks kept: the first grid axis is
indexed by the x range (:439); an image contributes nothing when it has no usable box or when its LAST
box is None (:450-455)."""
from __future__ import annotations

from ctypes import c_int, c_long
from typing import List

import torch

from ._lib import check, lib, ptr, stream
from .functional import logits_from_features
from .utils_image import patch_from_norm_bbox


class _BBoxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, boxes):
        grid = grid.float()
        if grid.stride(-1) != 1:
            grid = grid.contiguous()
        nbox, E = boxes.shape[0], grid.shape[-1]
        out = torch.empty(nbox, E, dtype=torch.float32, device=grid.device)
        check(lib().ce_bbox_pool_fwd(ptr(grid), c_long(grid.stride(0)), c_long(grid.stride(1)), c_long(grid.stride(2)),
                                     ptr(boxes), ptr(out), c_int(nbox), c_int(E), stream()), "ce_bbox_pool_fwd")
        ctx.saved, ctx.shape = boxes, tuple(grid.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        boxes = ctx.saved
        B, g, _, E = ctx.shape
        dgrid = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        dout = dout.contiguous().float()
        check(lib().ce_bbox_pool_bwd(ptr(dout), ptr(boxes), ptr(dgrid), c_int(g), c_int(boxes.shape[0]), c_int(E), stream()),
              "ce_bbox_pool_bwd")
        return dgrid, None


def region_losses(model, grid_features, bboxs, bbox_desc_vec, bbox_label_vec, train_arg: str):
    dev = grid_features.device
    pn = model.visual.patch_num
    use_label = train_arg.startswith("desc_type")
    use_role_text = train_arg.startswith("desc_type_text")
    box_rows: List[List[int]] = []
    descs, labs, groups = [], [], []
    for image_idx, bbox_image in enumerate(bboxs):
        start = len(box_rows)
        last = None
        for bbox_id, bbox in enumerate(bbox_image):
            last = bbox
            if bbox is None:
                continue
            x0, y0, x1, y1 = patch_from_norm_bbox(bbox, patch_size=pn)
            # python slicing clamps to the grid; an empty slice keeps the reference's mean-of-nothing = NaN
            x0, y0, x1, y1 = max(0, min(x0, pn)), max(0, min(y0, pn)), max(0, min(x1, pn)), max(0, min(y1, pn))
            box_rows.append([image_idx, x0, y0, max(x1, x0), max(y1, y0)])
            descs.append(bbox_desc_vec[image_idx][bbox_id])
            if use_label:
                labs.append(bbox_label_vec[image_idx][bbox_id])
        n = len(box_rows) - start
        if n == 0 or last is None:      # model_clip.py:450-455
            del box_rows[start:]
            del descs[start:]
            if use_label:
                del labs[start:]
            continue
        groups.append((start, n))
    zero = torch.zeros((), dtype=torch.float32, device=dev)
    if not groups:
        return zero, zero
    boxes = torch.tensor(box_rows, dtype=torch.int32, device=dev)
    region = _BBoxPoolFn.apply(grid_features, boxes)                                  # [nbox, E]
    desc_f = model.encode_text(torch.stack(descs).to(dev))                            # one tower pass for all roles
    lab_f = model.encode_text(torch.stack(labs).to(dev)) if use_label else None
    loss_per_bbox, loss_per_arg = zero, zero
    for start, n in groups:
        r = region[start:start + n]
        d = desc_f[start:start + n]
        y = torch.arange(n, device=dev)
        lpb, lpa = logits_from_features(r, d, model.logit_scale, True)               # s r d^T , s d r^T
        loss_per_bbox = loss_per_bbox + model.loss_func(lpb, y)
        loss_per_arg = loss_per_arg + model.loss_func(lpa, y)
        if use_label:
            l = lab_f[start:start + n]
            lpb2, lpa2 = logits_from_features(r, l, model.logit_scale, True)
            loss_per_bbox = loss_per_bbox + model.loss_func(lpb2, y)
            loss_per_arg = loss_per_arg + model.loss_func(lpa2, y)
            if use_role_text:
                lpr, _ = logits_from_features(d, l, model.logit_scale, True, want="image")   # s d l^T
                loss_per_arg = loss_per_arg + model.loss_func(lpr, y)
    return loss_per_bbox, loss_per_arg
