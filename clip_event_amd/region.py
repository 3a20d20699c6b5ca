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


class _RegionNCEFn(torch.autograd.Function):
    """(loss_per_bbox, loss_per_arg) of model_clip.py:456-488 over every image group (``ce_region_nce_fwd/bwd``)."""

    @staticmethod
    def forward(ctx, region, desc_f, lab_f, logit_scale, offsets, groups, max_rows, use_label, role_text):
        region, desc_f = region.contiguous().float(), desc_f.contiguous().float()
        lab_f = lab_f.contiguous().float() if lab_f is not None else None
        ls = logit_scale.detach().reshape(1)
        out = torch.zeros(2, dtype=torch.float32, device=region.device)
        check(lib().ce_region_nce_fwd(ptr(region), ptr(desc_f), ptr(lab_f), ptr(offsets), c_int(groups), c_int(max_rows),
                                      c_int(region.shape[1]), ptr(ls), c_int(1 if use_label else 0),
                                      c_int(1 if role_text else 0), ptr(out[0:1]), ptr(out[1:2]), stream()),
              "ce_region_nce_fwd")
        ctx.saved = (region, desc_f, lab_f, ls, offsets)
        ctx.cfg = (groups, max_rows, use_label, role_text)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_bbox, g_arg):
        region, desc_f, lab_f, ls, offsets = ctx.saved
        groups, max_rows, use_label, role_text = ctx.cfg
        dev = region.device
        g = torch.zeros(2, dtype=torch.float32, device=dev)
        if g_bbox is not None:
            g[0] = g_bbox
        if g_arg is not None:
            g[1] = g_arg
        dr, dd = torch.zeros_like(region), torch.zeros_like(desc_f)
        dl = torch.zeros_like(lab_f) if lab_f is not None else None
        dls = torch.zeros(1, dtype=torch.float32, device=dev)
        check(lib().ce_region_nce_bwd(ptr(region), ptr(desc_f), ptr(lab_f), ptr(offsets), c_int(groups), c_int(max_rows),
                                      c_int(region.shape[1]), ptr(ls), c_int(1 if use_label else 0),
                                      c_int(1 if role_text else 0), ptr(g[0:1]), ptr(g[1:2]), ptr(dr), ptr(dd), ptr(dl),
                                      ptr(dls), stream()), "ce_region_nce_bwd")
        return dr, dd, dl, dls.reshape(()), None, None, None, None, None


def _stack_rows(rows, lens, dev):
    """The role / label token rows of the usable boxes as one [n, T] matrix on the GPU.  Rows that are still on the host
    (a data loader's CPU tensors) are stacked there, their lengths taken on the host, and copied once; rows on the GPU
    carry their host-side lengths along when their per-image matrices were tagged (functional.attach_lengths)."""
    from .functional import attach_lengths, tokens_to_device
    if all(not r.is_cuda for r in rows):
        return tokens_to_device(torch.stack(rows), dev)
    out = torch.stack([r.to(dev) for r in rows])
    if all(l is not None for l in lens):
        attach_lengths(out, lens)
    return out


class RegionPlan:
    """Host-side layout of one batch's region branch (model_clip.py:430-455): the pooled boxes, the role / label token
    matrices of the usable boxes, and the per-image groups."""
    __slots__ = ("boxes", "descs", "labs", "offsets", "groups", "max_rows", "use_label", "use_role_text")


def region_plan(model, bboxs, bbox_desc_vec, bbox_label_vec, train_arg: str, dev):
    """Everything of the region branch that does not need a tower: ``None`` when no image has a usable box."""
    pn = model.visual.patch_num
    use_label = train_arg.startswith("desc_type")
    use_role_text = train_arg.startswith("desc_type_text")
    box_rows: List[List[int]] = []
    descs, labs, dlens, llens, groups = [], [], [], [], []

    def row_len(mat, i):
        l = getattr(mat, "_ce_lengths", None)
        return None if l is None else int(l[i])

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
            dlens.append(row_len(bbox_desc_vec[image_idx], bbox_id))
            if use_label:
                labs.append(bbox_label_vec[image_idx][bbox_id])
                llens.append(row_len(bbox_label_vec[image_idx], bbox_id))
        n = len(box_rows) - start
        if n == 0 or last is None:      # model_clip.py:450-455
            del box_rows[start:]
            del descs[start:]
            del dlens[start:]
            if use_label:
                del labs[start:]
                del llens[start:]
            continue
        groups.append((start, n))
    if not groups:
        return None
    plan = RegionPlan()
    plan.boxes = torch.tensor(box_rows, dtype=torch.int32, device=dev)
    plan.descs = _stack_rows(descs, dlens, dev)
    plan.labs = _stack_rows(labs, llens, dev) if use_label else None
    plan.offsets = torch.tensor([g[0] for g in groups] + [groups[-1][0] + groups[-1][1]], dtype=torch.int32, device=dev)
    plan.groups = len(groups)
    plan.max_rows = max(n for _, n in groups)
    if plan.max_rows > 16:
        raise RuntimeError(f"{plan.max_rows} boxes in one image: the region InfoNCE kernel takes at most 16")
    plan.use_label, plan.use_role_text = use_label, use_role_text
    return plan


def region_losses_from_features(model, grid_features, plan: RegionPlan, desc_f, lab_f):
    """(loss_per_bbox, loss_per_arg) from the image grid and the role / label text features of ``plan``'s rows: one pooling
    launch, then all images' n x n InfoNCE terms, both directions and the label / role-text variants, in one launch (and one
    for the backward): launch count independent of the batch size (the reference loops over images, :456-488)."""
    region = _BBoxPoolFn.apply(grid_features, plan.boxes)                             # [nbox, E]
    return _RegionNCEFn.apply(region, desc_f, lab_f, model.logit_scale, plan.offsets, plan.groups, plan.max_rows,
                              plan.use_label, plan.use_role_text)


def region_losses(model, grid_features, bboxs, bbox_desc_vec, bbox_label_vec, train_arg: str):
    dev = grid_features.device
    plan = region_plan(model, bboxs, bbox_desc_vec, bbox_label_vec, train_arg, dev)
    if plan is None:
        zero = torch.zeros((), dtype=torch.float32, device=dev)
        return zero, zero
    desc_f = model.encode_text(plan.descs)                                            # one tower pass for all roles
    lab_f = model.encode_text(plan.labs) if plan.use_label else None
    return region_losses_from_features(model, grid_features, plan, desc_f, lab_f)
