"""Seeded synthetic inputs of the shapes BASELINE.json names.

numpy ``Generator`` streams (PCG64) are used throughout so that the same seed
gives the same bytes on the build container and on the GPU box.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

ASCII_CAPTIONS: Tuple[str, ...] = (
    "a photo of a cat",
    "Protesters march through the streets of the capital on Saturday.",
    "A soldier is transported to a hospital after an attack near the border.",
    "The president speaks to reporters at the White House in Washington.",
    "Rescue workers search for survivors after the earthquake destroyed buildings.",
    "Police officers arrest a man during a demonstration in the city centre.",
    "Voters cast their ballots at a polling station during the election.",
    "Firefighters try to extinguish a fire at a warehouse.",
)


def synthetic_images(batch: int, resolution: int = 224, seed: int = 999) -> torch.Tensor:
    """N(0,1) images [B,3,R,R] fp32: the statistics after ``Normalize`` (clip.py:62-69)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((batch, 3, resolution, resolution), dtype=np.float32))


def synthetic_tokens(n: int, context_length: int = 77, vocab_size: int = 49408, seed: int = 999,
                     min_len: int = 8, max_len: Optional[int] = None) -> torch.Tensor:
    """Token rows as ``clip.tokenize`` lays them out (clip.py:187-199): SOT =
    vocab-2, ``len`` ids in [1, vocab-3], EOT = vocab-1 (the row maximum, which
    ``argmax`` relies on, model_clip.py:415), zero padding."""
    rng = np.random.default_rng(seed)
    max_len = context_length - 2 if max_len is None else max_len
    min_len = min(min_len, max_len)
    out = np.zeros((n, context_length), dtype=np.int64)
    lens = rng.integers(min_len, max_len + 1, size=n)
    for i, ln in enumerate(lens):
        out[i, 0] = vocab_size - 2
        out[i, 1:1 + ln] = rng.integers(1, vocab_size - 2, size=ln)
        out[i, 1 + ln] = vocab_size - 1
    return torch.from_numpy(out)


def synthetic_bboxes(batch: int, seed: int = 999, max_roles: int = 4, none_frac: float = 0.25):
    """Per image a list of 1..max_roles boxes ``(x0,y0,x1,y1)`` in [0,1] or ``None``
    (input format of the ``train_arg`` branch, dataset_sr.py:159-168)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(batch):
        n = int(rng.integers(1, max_roles + 1))
        boxes = []
        for _ in range(n):
            if rng.random() < none_frac:
                boxes.append(None)
                continue
            xs = np.sort(rng.random(2))
            ys = np.sort(rng.random(2))
            # keep boxes non-degenerate: at least a sliver wide
            boxes.append((float(xs[0]), float(ys[0]), float(min(1.0, xs[1] + 1e-3)), float(min(1.0, ys[1] + 1e-3))))
        out.append(boxes)
    return out


# CLIP constructor arguments (model_clip.py:267-286) of the geometries the benches use
GEOMETRY = {
    "vit_b32": (512, 224, 12, 768, 32, 77, 49408, 512, 8, 12),
    "vit_b16": (512, 224, 12, 768, 16, 77, 49408, 512, 8, 12),
    "vit_l14_336": (768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12),
}


def synthetic_model(geometry: str = "vit_b32", seed: int = 0):
    """Random-init ``CLIP`` of a named geometry with the reference's own initialisation
    (``CLIP.initialize_parameters``, model_clip.py:348-375) under a fixed CPU seed: the benches' weights (there is
    no network for checkpoints).  The same weights on every rank and every box."""
    from .model import CLIP
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = CLIP(*GEOMETRY[geometry])
    finally:
        torch.random.set_rng_state(state)
    return model


def synthetic_entities(batch: int, resolution: int = 224, context_length: int = 77, vocab_size: int = 49408,
                       seed: int = 999, max_objects: int = 6, max_entities: int = 10):
    """Inputs of ``sim_entity`` / ``CriterionAlignment`` (engine.py:57-63) as SURVEY 8(d) c4 lays them out: per
    image 1 + U[0, max_objects] object crops (slot 0 = the whole image, model_clip.py:686) and U[1, max_entities]
    entity mentions, padded to the batch maxima with 0/1 masks.  Returns ``(object_vec [B,O,3,R,R] f32, object_num
    [B,O] int64, entitytxt_vec [B,M,T] int64, entitytxt_num [B,M] int64)``; padding crops are zeros, padding
    mentions are the empty caption (SOT EOT)."""
    rng = np.random.default_rng(seed)
    n_obj = 1 + rng.integers(0, max_objects + 1, size=batch)
    n_ent = rng.integers(1, max_entities + 1, size=batch)
    O, M = int(n_obj.max()), int(n_ent.max())
    obj = np.zeros((batch, O, 3, resolution, resolution), dtype=np.float32)
    obj_num = np.zeros((batch, O), dtype=np.int64)
    ent = np.zeros((batch, M, context_length), dtype=np.int64)
    ent[:, :, 0] = vocab_size - 2
    ent[:, :, 1] = vocab_size - 1
    ent_num = np.zeros((batch, M), dtype=np.int64)
    for b in range(batch):
        obj[b, :n_obj[b]] = rng.standard_normal((n_obj[b], 3, resolution, resolution), dtype=np.float32)
        obj_num[b, :n_obj[b]] = 1
        toks = synthetic_tokens(int(n_ent[b]), context_length, vocab_size, seed=int(rng.integers(1 << 30)), min_len=1,
                                max_len=min(6, context_length - 2)).numpy()
        ent[b, :n_ent[b]] = toks
        ent_num[b, :n_ent[b]] = 1
    return torch.from_numpy(obj), torch.from_numpy(obj_num), torch.from_numpy(ent), torch.from_numpy(ent_num)


def synthetic_role_texts(bboxs, context_length: int = 77, vocab_size: int = 49408, seed: int = 999, max_len: int = 12):
    """One token row per box of every image (``bbox_desc_vec`` / ``bbox_label_vec`` of model_clip.py:419):
    list[B] of int64 [n_roles, T]."""
    rng = np.random.default_rng(seed)
    return [synthetic_tokens(len(b), context_length, vocab_size, seed=int(rng.integers(1 << 30)), min_len=1,
                             max_len=min(max_len, context_length - 2)) for b in bboxs]
