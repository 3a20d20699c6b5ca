"""Byte-level BPE tokenizer: drop-in for ``clip.tokenize`` (reference
``clip.py:168-201``) and ``SimpleTokenizer`` (``model_simple_tokenizer.py:62-132``).

Integer, bit-exact contract: vocabulary = 256 byte symbols,
the same 256 with the end-of-word marker, 48,894 merges in rank order, then
``<|startoftext|>`` (49406) and ``<|endoftext|>`` (49407); ids are zero-padded to
``context_length`` and over-long inputs are truncated with EOT forced into the
last column (clip.py:193-196).

The merge list is the OpenAI CLIP BPE data file (first 48,894 merges; data, not
code) shipped as ``assets/bpe_merges_48894.txt.gz``.  Host-side Python; the
synthetic benchmark pre-tokenises, so this is outside GPU timing.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import regex
import torch

try:  # ftfy only repairs mojibake; absent from this image -> identity (exact for clean text)
    from ftfy import fix_text as _fix_text
except Exception:  # pragma: no cover - depends on the image
    def _fix_text(s: str) -> str:
        return s

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "bpe_merges_48894.txt.gz")
_EOW = "</w>"
_SOT, _EOT = "<|startoftext|>", "<|endoftext|>"
_N_MERGES = 49152 - 256 - 2   # model_simple_tokenizer.py:67


@lru_cache()
def byte_symbols() -> Dict[int, str]:
    """Printable stand-in character for each of the 256 byte values
    (model_simple_tokenizer.py:15-35): printable latin-1 bytes map to
    themselves, the remaining 68 bytes to code points 256.. in byte order."""
    keep = list(range(0x21, 0x7F)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table = {b: chr(b) for b in keep}
    nxt = 256
    for b in range(256):
        if b not in table:
            table[b] = chr(nxt)
            nxt += 1
    # the reference orders its vocabulary by the `keep` list first, then the remapped bytes
    ordered = {b: table[b] for b in keep}
    for b in range(256):
        if b not in ordered:
            ordered[b] = table[b]
    return ordered


class BPETokenizer:
    def __init__(self, merges_path: str = _ASSET):
        with gzip.open(merges_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        if lines and "#version" in lines[0]:
            lines = lines[1:]              # accept the original file with its header line too
        pairs = [tuple(l.split()) for l in lines[:_N_MERGES]]
        if len(pairs) != _N_MERGES or any(len(p) != 2 for p in pairs):
            raise RuntimeError(f"BPE merge file {merges_path} does not hold {_N_MERGES} merges")
        self.byte_encoder = byte_symbols()
        self.byte_decoder = {c: b for b, c in self.byte_encoder.items()}
        symbols = list(self.byte_encoder.values())
        vocab = symbols + [s + _EOW for s in symbols] + [a + b for a, b in pairs] + [_SOT, _EOT]
        self.encoder: Dict[str, int] = {tok: i for i, tok in enumerate(vocab)}
        self.decoder: Dict[int, str] = {i: tok for tok, i in self.encoder.items()}
        self.rank: Dict[Tuple[str, str], int] = {p: i for i, p in enumerate(pairs)}
        self._cache: Dict[str, Tuple[str, ...]] = {_SOT: (_SOT,), _EOT: (_EOT,)}
        # model_simple_tokenizer.py:78
        self._splitter = regex.compile(
            r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
            regex.IGNORECASE)

    @property
    def sot_token(self) -> int:
        return self.encoder[_SOT]

    @property
    def eot_token(self) -> int:
        return self.encoder[_EOT]

    def _merge_word(self, token: str) -> Tuple[str, ...]:
        """Greedy lowest-rank-first pair merging (model_simple_tokenizer.py:80-119)."""
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        parts: List[str] = list(token[:-1]) + [token[-1] + _EOW]
        while len(parts) > 1:
            best_rank, best = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best_rank, best = r, (a, b)
            if best is None:
                break
            a, b = best
            merged: List[str] = []
            i = 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        out = tuple(parts)
        self._cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        """Clean (ftfy, double html-unescape, strip, collapse whitespace), lower,
        split, byte-map, merge (model_simple_tokenizer.py:50-59, :121-127)."""
        text = html.unescape(html.unescape(_fix_text(text))).strip()
        text = regex.sub(r"\s+", " ", text).strip().lower()
        ids: List[int] = []
        for piece in self._splitter.findall(text):
            mapped = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._merge_word(mapped))
        return ids

    def decode(self, tokens: Iterable[int]) -> str:
        """model_simple_tokenizer.py:129-132."""
        text = "".join(self.decoder[int(t)] for t in tokens)
        raw = bytearray(self.byte_decoder[c] for c in text)
        return raw.decode("utf-8", errors="replace").replace(_EOW, " ")


@lru_cache()
def default_tokenizer() -> BPETokenizer:
    return BPETokenizer()


def tokenize(texts: Union[str, Sequence[str]], context_length: int = 77) -> torch.LongTensor:
    """``clip.tokenize`` (clip.py:168-201): ``[n, context_length]`` int64 (on the host, as in the reference)."""
    if isinstance(texts, str):
        texts = [texts]
    tk = default_tokenizer()
    sot, eot = tk.sot_token, tk.eot_token
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, text in enumerate(texts):
        ids = [sot] + tk.encode(text) + [eot]
        if len(ids) > context_length:
            ids = ids[:context_length]
            ids[-1] = eot
        out[i, :len(ids)] = torch.tensor(ids, dtype=torch.long)
    # host-side caption lengths (argmax + 1, the EOT being the largest id: model_clip.py:415) ride along as a tag, so the
    # text tower of the HIP path never has to read them back from the GPU (functional.text_packing / tokens_to_device)
    out._ce_lengths = out.numpy().argmax(axis=-1).astype("int64") + 1
    return out
