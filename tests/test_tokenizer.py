"""CPU, integer, bit-exact: the product tokenizer against ids captured from the reference's
``clip.tokenize`` (ftfy stubbed as identity; cases are ASCII so that is exact)."""
import numpy as np
import torch

from clip_event_amd.tokenizer import tokenize, default_tokenizer
from tests.util import golden_json, golden_npz

G = golden_json()["tokenizer"]
Z = golden_npz("tokenizer.npz")


def test_known_answer():
    ids = tokenize("a photo of a cat")
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (1, 77)
    assert ids[0, :7].tolist() == [49406, 320, 1125, 539, 320, 2368, 49407]
    assert ids[0, 7:].sum().item() == 0


def test_vocab_layout():
    tk = default_tokenizer()
    assert len(tk.encoder) == G["vocab_size"] == 49408
    assert tk.sot_token == G["sot"] == 49406 and tk.eot_token == G["eot"] == 49407


def test_ids_bit_exact_ctx77_and_ctx20():
    assert np.array_equal(tokenize(G["cases"]).numpy(), Z["ids77"])
    assert np.array_equal(tokenize(G["cases"], context_length=20).numpy(), Z["ids20"])


def test_truncation_forces_eot_and_argmax():
    ids = tokenize(G["cases"])
    assert ids.argmax(-1).tolist() == G["argmax77"]
    long_row = tokenize("x " * 200)[0]
    assert long_row[76].item() == 49407 and long_row.argmax().item() == 76
    assert tokenize("")[0, :2].tolist() == [49406, 49407]


def test_decode_roundtrip():
    tk = default_tokenizer()
    ids = tokenize(G["cases"][1])[0]
    assert tk.decode(ids[1:int(ids.argmax())].tolist()) == G["decode0"]
