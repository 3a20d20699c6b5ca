"""A synthetic "trained-like" parameter state and caption batch for the fp16-stream tests.

Every fixture of the suite is random-init (the reference ships no checkpoint), where activations are O(1).  A
trained CLIP is not like that: a handful of residual channels carry values in the hundreds ("massive activations"),
LayerNorm gains span an order of magnitude, and ``logit_scale`` sits at its ceiling ln 100 (model_clip.py:330,502 never clamp
it).  ``hostile_state`` rewrites a seeded random-init state dict into that regime with plain arithmetic on the tensors, so
the CPU oracle (fp32 = the reference's arithmetic) and the HIP path see the same numbers:

* ``n_outlier`` residual channels per tower receive a constant of magnitude U[100, 300] through the bias of block 0's
  ``mlp.c_proj`` (every token then carries it down the whole tower) and, in the text tower, through the positional
  embedding as well; the rows of later ``c_proj`` weights that write those channels are scaled x4;
* LayerNorm gains: 6 % of the channels x U[3, 10], the outlier channels x 0.05 in the blocks' and the final LayerNorms (trained
  models squash them there);
* ``logit_scale = ln 100``.

``hostile_tokens`` builds captions of the two extreme lengths the packed text tower has to handle (3 tokens = SOT, one
id, EOT; and the full context) beside ordinary ones."""
import math

import numpy as np
import torch


def hostile_state(sd, cfg, seed: int = 0, n_outlier: int = 3, magnitude=(100.0, 300.0), row_scale: float = 4.0):
    rng = np.random.default_rng(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    for prefix, width, layers, pos in (("visual.transformer.", cfg.vision_width, cfg.vision_layers, None),
                                       ("transformer.", cfg.transformer_width, cfg.transformer_layers, "positional_embedding")):
        ch = rng.choice(width, size=n_outlier, replace=False)
        amp = rng.uniform(*magnitude, size=n_outlier) * rng.choice([-1.0, 1.0], size=n_outlier)
        sd[prefix + "resblocks.0.mlp.c_proj.bias"][ch] += torch.from_numpy(amp).float()
        if pos is not None:
            sd[pos][:, ch] += torch.from_numpy(0.5 * amp).float()
        for l in range(1, layers):
            sd[prefix + f"resblocks.{l}.mlp.c_proj.weight"][ch, :] *= row_scale
        for l in range(layers):
            for ln in ("ln_1", "ln_2"):
                w = sd[prefix + f"resblocks.{l}.{ln}.weight"]
                big = rng.choice(width, size=max(1, int(0.06 * width)), replace=False)
                w[big] *= torch.from_numpy(rng.uniform(3.0, 10.0, size=len(big))).float()
                w[ch] *= 0.05
        final = "visual.ln_post.weight" if pos is None else "ln_final.weight"
        sd[final][ch] *= 0.05
    sd["logit_scale"] = torch.tensor(math.log(100.0))
    return sd


def hostile_tokens(n: int, context_length: int, vocab_size: int, seed: int = 0) -> torch.Tensor:
    """[n, T] int64 in ``clip.tokenize``'s layout: row 0 has 3 tokens, row 1 fills the context, the rest U[3, T]."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, context_length), dtype=np.int64)
    for i in range(n):
        body = 1 if i == 0 else (context_length - 2 if i == 1 else int(rng.integers(1, context_length - 1)))
        out[i, 0] = vocab_size - 2
        out[i, 1:1 + body] = rng.integers(1, vocab_size - 2, size=body)
        out[i, 1 + body] = vocab_size - 1
    return torch.from_numpy(out)
