"""Checkpoint interchange with the reference.

Same file layout as ``save_model_on_master`` (engine.py:202-218): ``{'epoch', 'model', 'state_dict', 'perf',
'optimizer'}`` written with ``torch.save`` as ``<task>_<epoch>.pth``; the ``state_dict`` has the reference's keys
and shapes, ``optimizer`` is ``torch.optim.Adam``'s format (optim.FusedAdam speaks it), so files move both ways.
Loading follows train.py:101-124 (resume) and also accepts a bare state dict such as the OpenAI ViT-B/32 weights
(fp16 tensors are widened to the fp32 masters by ``build_model``).  Files are read with ``weights_only=True``:
nothing in a checkpoint is executed.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Optional, Tuple

import torch

from . import distributed as D
from .model import build_model

CKPT_KEYS = ("epoch", "model", "state_dict", "perf", "optimizer")


def checkpoint_dict(model, optimizer, task: str, epoch: int, best_perf: float) -> Dict[str, Any]:
    states = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = optimizer.state_dict() if optimizer is not None else None
    if opt is not None:
        opt = {"state": {i: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
                         for i, st in opt["state"].items()},
               "param_groups": opt["param_groups"]}
    return {"epoch": epoch, "model": task, "state_dict": states, "perf": best_perf, "optimizer": opt}


def save_model_on_master(model, ckpt_dir: str, task: str, epoch: int, best_perf: float, optimizer=None) -> Optional[str]:
    """engine.py:202-218: rank 0 writes ``<ckpt_dir>/<task>_<epoch>.pth``; other ranks return None."""
    if D.rank() != 0:
        return None
    path = os.path.join(ckpt_dir, "%s_%s.pth" % (task, epoch))
    logging.info("=> saving checkpoint to {}".format(ckpt_dir))
    try:
        torch.save(checkpoint_dict(model, optimizer, task, epoch, best_perf), path)
    except Exception:
        logging.error("=> error when saving checkpoint!")
        return None
    return path


def load_checkpoint(path: str, device=None, is_train: bool = True) -> Tuple[Any, Optional[dict], int, float]:
    """train.py:101-124: returns ``(model, optimizer_state, begin_epoch, best_perf)``.  ``path`` holds either the
    reference's checkpoint dictionary or a bare CLIP state dict (then optimizer_state is None, epoch 0, perf 0)."""
    if not os.path.exists(path):
        raise FileNotFoundError("=> error when loading checkpoint (cannot find checkpoint): {}".format(path))
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(blob, dict) and "state_dict" in blob:
        state_dict = blob["state_dict"]
        opt_state = blob.get("optimizer")
        begin_epoch = blob.get("epoch" if is_train else "step", 0)
        best_perf = blob.get("perf", 0.0)
    else:
        state_dict, opt_state, begin_epoch, best_perf = blob, None, 0, 0.0
    model = build_model(dict(state_dict))
    if device is not None:
        model = model.to(device)
    return model, opt_state, begin_epoch, best_perf
