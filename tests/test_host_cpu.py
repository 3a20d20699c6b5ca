"""CPU (no GPU): host logic of the product, the C-ABI library's exports, and the loud failure
when the HIP path cannot run."""
import ctypes
import os

import numpy as np
import re

import pytest
import torch

from oracle import clip_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from clip_event_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "clip_event_hip.h")).read()
    names = sorted(set(re.findall(r"\b(ce_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.ce_version.restype = ctypes.c_int
    assert lib.ce_version() >= 1


def test_argument_errors_are_reported_without_a_gpu():
    """Argument validation happens before any launch: returns -EINVAL and a message."""
    from clip_event_amd._lib import lib
    cl = lib()
    rc = cl.ce_gemm_nt(None, ctypes.c_long(8), None, ctypes.c_long(8), 0, 0, 0, 0, None, None, ctypes.c_long(0), None,
                       ctypes.c_long(0), None, ctypes.c_long(0), None, ctypes.c_long(0), None)
    assert rc == -22
    assert b"empty problem" in cl.ce_last_error()


def test_state_dict_keys_and_build_model_match_reference_layout():
    from clip_event_amd.model import CLIP, build_model
    cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2)
    sd = O.init_params(cfg, 3)
    m = build_model({k: v.clone() for k, v in sd.items()})
    assert list(m.state_dict().keys()) == list(O.param_shapes(cfg).keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(O.param_shapes(cfg)[k]) and torch.equal(v, sd[k])
    assert m.training and m.visual.input_resolution == 64 and m.visual.patch_num == 2
    assert m.dtype == torch.float32 and m.context_length == 20 and m.vocab_size == 512
    m.set_hyps(constrastive_overbatch=False, alignment=True, multiattention=True)
    assert (m.constrastive_overbatch, m.alignment, m.multiattention) == (False, True, True)
    n_params = sum(p.numel() for p in CLIP(512, 224, 12, 768, 32, 77, 49408, 512, 8, 12).parameters())
    assert n_params == 151277313                      # ViT-B/32


def test_no_cpu_fallback():
    from clip_event_amd.model import build_model
    from clip_event_amd._lib import HipExtensionMissing
    from clip_event_amd.losses import CriterionContrastive
    cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2)
    m = build_model(O.init_params(cfg, 3))
    with pytest.raises(HipExtensionMissing):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 20, dtype=torch.long))
    with pytest.raises(RuntimeError):
        CriterionContrastive("ce")(torch.zeros(2, 2), torch.zeros(2, 2), None, None, index_pos=torch.arange(2))
    with pytest.raises(RuntimeError):
        CriterionContrastive("hinge")


def test_global_labels_match_single_process_layout():
    from clip_event_amd.distributed import global_labels
    for overbatch in (True, False):
        yi, yt, ip = global_labels(4, 1, 2, overbatch, rank_=0)
        ri, rt, rp = O.build_labels(4, 1, 2, overbatch)
        assert torch.equal(yi, ri) and torch.equal(yt, rt) and torch.equal(ip, rp)
    # rank 1 of W=2, B=3, K=2: images 3..5, positive columns (3..5)*2, local positive rows 0,2,4
    yi, yt, ip = global_labels(3, 1, 1, True, rank_=1)
    assert yi.tolist() == [6, 8, 10] and yt.tolist() == [3, 3, 4, 4, 5, 5] and ip.tolist() == [0, 2, 4]


def test_patch_from_norm_bbox_integer_contract():
    from clip_event_amd import patch_from_norm_bbox
    from tests.util import golden_json
    assert patch_from_norm_bbox((0.1, 0.1, 0.6, 0.7), 7) == (0, 0, 5, 5)       # SURVEY.md R2 [probed]
    for case in golden_json()["region"]["bbox7"]:
        assert list(patch_from_norm_bbox(tuple(case["bbox"]), 7)) == case["idx"]


def test_lr_schedulers_match_reference_sequence():
    """Product schedulers (host arithmetic, optim.py) driven as engine.py:97 does, against the reference's lr
    sequences and the oracle."""
    import torch
    from oracle import clip_oracle as O
    from clip_event_amd.optim import WarmupCosineLR, WarmupMultiStepLR, build_lr_scheduler
    from tests.util import golden_json
    G = golden_json()["sched"]

    def run(make, n):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([w], lr=G["base_lr"])
        sch = make(opt)
        out = []
        for _ in range(n):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        return out

    c = G["cosine"]
    got = run(lambda o: WarmupCosineLR(o, c["max_iters"], warmup_epochs=c["warmup_epochs"]), c["n"])
    np.testing.assert_allclose(got, c["lr"], rtol=1e-12)
    np.testing.assert_allclose(got, [O.lr_warmup_cosine(G["base_lr"], i, c["max_iters"], warmup_epochs=c["warmup_epochs"])
                                     for i in range(c["n"])], rtol=1e-12)
    c = G["cosine_const"]
    got = run(lambda o: WarmupCosineLR(o, c["max_iters"], warmup_factor=c["warmup_factor"], warmup_epochs=c["warmup_epochs"],
                                       warmup_method="constant"), c["n"])
    np.testing.assert_allclose(got, c["lr"], rtol=1e-12)
    c = G["multistep"]
    got = run(lambda o: WarmupMultiStepLR(o, c["milestones"], gamma=c["gamma"], warmup_epochs=c["warmup_epochs"]), c["n"])
    np.testing.assert_allclose(got, c["lr"], rtol=1e-12)
    # build_lr_scheduler (engine.py:154-176): kinds and the error text
    w = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([w], lr=1e-3)
    cfg = {"lr_scheduler": "warmup", "max_epoch": 10, "warmup_epoch": 2, "lr_steps": [3], "lr_gamma": 0.1}
    assert isinstance(build_lr_scheduler(cfg, opt, 0), WarmupCosineLR)
    assert build_lr_scheduler(dict(cfg, lr_scheduler="none"), opt, 0) is None
    assert isinstance(build_lr_scheduler(dict(cfg, lr_scheduler="multisteplr"), opt, 0), torch.optim.lr_scheduler.MultiStepLR)
    with pytest.raises(RuntimeError, match="Invalid lr scheduler"):
        build_lr_scheduler(dict(cfg, lr_scheduler="bogus"), opt, 0)
    with pytest.raises(ValueError, match="increasing"):
        WarmupMultiStepLR(opt, [5, 2])


def test_checkpoint_layout_round_trip(tmp_path):
    """engine.py:202-218 / train.py:101-124: same five keys, same state-dict keys; fp16 weights (OpenAI release
    format) widen to the fp32 masters; bare state dicts load too.  Read back with weights_only=True."""
    import torch
    from clip_event_amd import checkpoint as C
    from clip_event_amd.model import build_model
    cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2)
    sd = O.init_params(cfg, 4)
    m = build_model({k: v.clone() for k, v in sd.items()})
    path = C.save_model_on_master(m, str(tmp_path), "clipevent", 3, 0.25, optimizer=None)
    assert path is not None and path.endswith("clipevent_3.pth")
    blob = torch.load(path, map_location="cpu", weights_only=True)
    assert tuple(blob.keys()) == C.CKPT_KEYS and blob["epoch"] == 3 and blob["model"] == "clipevent" and blob["perf"] == 0.25
    assert list(blob["state_dict"].keys()) == list(sd.keys())
    m2, opt_state, epoch, perf = C.load_checkpoint(path)
    assert (opt_state, epoch, perf) == (None, 3, 0.25)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # bare fp16 state dict with the JIT archive's three extra keys
    half = {k: (v.half() if v.dtype == torch.float32 else v) for k, v in sd.items()}
    half.update(input_resolution=torch.tensor(64), context_length=torch.tensor(20), vocab_size=torch.tensor(512))
    bare = str(tmp_path / "bare.pt")
    torch.save(half, bare)
    m3, opt_state, epoch, perf = C.load_checkpoint(bare)
    assert (opt_state, epoch, perf) == (None, 0, 0.0)
    for k, v in m3.state_dict().items():
        assert v.dtype == torch.float32 and torch.equal(v, sd[k].half().float()), k
    with pytest.raises(FileNotFoundError, match="cannot find checkpoint"):
        C.load_checkpoint(str(tmp_path / "missing.pth"))


def test_clip_module_surface(tmp_path):
    """clip.py:72-166 surface: names, error texts, loading a local state dict / reference-layout checkpoint."""
    import torch
    from clip_event_amd import clip
    assert clip.available_models() == ["RN50", "RN101", "RN50x4", "ViT-B/32"]
    with pytest.raises(RuntimeError, match="would have to be downloaded"):
        clip.load("ViT-B/32", device="cpu")
    with pytest.raises(RuntimeError, match="not found; available models"):
        clip.load(str(tmp_path / "nope.pt"), device="cpu")
    cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2)
    sd = O.init_params(cfg, 2)
    bare = str(tmp_path / "sd.pt")
    torch.save(sd, bare)
    with pytest.warns(UserWarning, match="not a JIT archive"):
        m, pre = clip.load(bare, device="cpu")
    assert m.visual.input_resolution == 64 and callable(pre)
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    wrapped = str(tmp_path / "ck.pth")
    torch.save({"epoch": 1, "model": "t", "state_dict": sd, "perf": 0.0, "optimizer": None}, wrapped)
    m2, _ = clip.load(wrapped, device="cpu", jit=False)
    assert all(torch.equal(v, sd[k]) for k, v in m2.state_dict().items())
    ids = clip.tokenize(["a photo of a cat"])
    assert ids.shape == (1, 77) and ids[0, :7].tolist() == [49406, 320, 1125, 539, 320, 2368, 49407]


def _persist_tiles(tiles_m, tiles_n, G, R):
    """Python mirror of persist_walk / persist_coords (csrc/gemm_common.hpp): tile list of every workgroup."""
    total = tiles_m * tiles_n
    seen = []
    for bid in range(G):
        if R > 0 and G % 8 == 0:
            xcd, local, per = bid & 7, bid >> 3, G >> 3
            q, r = total >> 3, total & 7
            start, cnt = xcd * q + min(xcd, r), q + (1 if xcd < r else 0)
            first, step, count = start + local, per, (cnt - local + per - 1) // per
            owner = (start, start + cnt)
        else:
            xcd, local = bid & 7, bid >> 3
            q, r = G >> 3, G & 7
            xb = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + local
            first, step, count = xb, G, (total - xb + G - 1) // G
            owner = (0, total)
        assert count >= 1
        for t in range(count):
            tile = first + t * step
            assert owner[0] <= tile < owner[1]
            if R > 0:
                chunk_tiles = R * tiles_n
                c, within = divmod(tile, chunk_tiles)
                rc = min(R, tiles_m - c * R)
                tn, tm = divmod(within, rc)
                tm += c * R
            else:
                tm, tn = divmod(tile, tiles_n)
            assert 0 <= tm < tiles_m and 0 <= tn < tiles_n
            seen.append((tm, tn, bid & 7))
    return seen


def test_persistent_gemm_tile_walk_is_a_bijection():
    """The XCD-owned walk of the persistent NT GEMM visits every tile exactly once for full, ragged and tiny chunks, and
    keeps every row panel inside at most two XCDs (the L2-residency claim of gemm_common.hpp)."""
    for tiles_m, tiles_n in [(80, 12), (80, 9), (68, 6), (68, 8), (113, 6), (32, 9), (33, 8), (257, 1), (7, 40), (100, 24)]:
        total = tiles_m * tiles_n
        G = min(total, 256)
        for R in (0, max(1, tiles_m // 8), 3, 1, tiles_m):
            seen = _persist_tiles(tiles_m, tiles_n, G, R)
            assert len(seen) == total and len({(a, b) for a, b, _ in seen}) == total, (tiles_m, tiles_n, R)
        seen = _persist_tiles(tiles_m, tiles_n, G, max(1, tiles_m // 8))      # the launcher's default chunk height
        if G % 8 == 0 and tiles_m >= 8:
            owners = {}
            for tm, _, xcd in seen:
                owners.setdefault(tm, set()).add(xcd)
            assert max(len(v) for v in owners.values()) <= 2, (tiles_m, tiles_n)


def test_caption_lengths_travel_on_the_host():
    """Round 4: the text tower sizes its launches from HOST-side caption lengths (no device read-back): ``clip.tokenize``
    tags its result, ``tokens_to_device`` / ``attach_lengths`` / ``host_lengths`` agree with argmax + 1 (first occurrence of
    the row maximum, model_clip.py:415), and the packed-layout metadata built from them is what the device path would build."""
    import numpy as np
    from clip_event_amd import synthetic as S
    from clip_event_amd.clip import tokenize
    from clip_event_amd.functional import attach_lengths, host_lengths, tokens_to_device
    t = tokenize(["a photo of a cat", "hello", ""])
    assert np.array_equal(t._ce_lengths, (t.argmax(dim=-1) + 1).numpy())
    assert t._ce_lengths.tolist()[-1] == 2                      # the empty caption: SOT EOT
    txt = S.synthetic_tokens(37, 77, 49408, seed=3, min_len=1)
    lens = host_lengths(txt)
    assert np.array_equal(lens, (txt.argmax(dim=-1) + 1).numpy())
    same = tokens_to_device(txt, "cpu")                         # already "there": returned as is (is_cuda False -> a copy with the tag)
    assert np.array_equal(getattr(same, "_ce_lengths"), lens)
    v = attach_lengths(txt.clone(), lens.tolist())
    assert v._ce_lengths.dtype == np.int64 and v._ce_lengths.shape == (37,)
    # a 3-D entity tensor: lengths are per row of the flattened [B * M, T] view
    ent = txt[:36].reshape(6, 6, 77)
    assert np.array_equal(host_lengths(ent), lens[:36])


def test_region_plan_keeps_the_reference_quirks():
    """``region.region_plan`` (the tower-free half of model_clip.py:430-455) against the oracle's loop: which boxes are pooled,
    which images are dropped (no usable box, or the LAST box None), the x / y transposition of the slice, the order of the
    role rows."""
    import torch
    from clip_event_amd import synthetic as S
    from clip_event_amd.region import region_plan
    from oracle import clip_oracle as O

    class _V:
        patch_num = 7

    class _M:
        visual = _V()

    boxes = S.synthetic_bboxes(24, seed=5, max_roles=4, none_frac=0.4)
    boxes[3] = [None, None]                       # nothing usable
    boxes[4] = [(0.1, 0.2, 0.5, 0.9), None]       # last box None: dropped although one box is usable
    boxes[5] = [None, (0.0, 0.0, 1.0, 1.0)]
    desc = S.synthetic_role_texts(boxes, seed=6)
    lab = S.synthetic_role_texts(boxes, seed=7)
    plan = region_plan(_M(), boxes, desc, lab, "desc_type_text", torch.device("cpu"))
    want_boxes, want_rows, want_groups = [], [], []
    for i, bx in enumerate(boxes):
        use = [(j, b) for j, b in enumerate(bx) if b is not None]
        if not use or bx[-1] is None:
            continue
        want_groups.append(len(use))
        for j, b in use:
            x0, y0, x1, y1 = O.patch_from_norm_bbox(b, 7)
            want_boxes.append([i, x0, y0, x1, y1])
            want_rows.append(desc[i][j])
    assert plan.boxes.tolist() == want_boxes
    assert torch.equal(plan.descs, torch.stack(want_rows))
    off = plan.offsets.tolist()
    assert [off[k + 1] - off[k] for k in range(len(off) - 1)] == want_groups and plan.groups == len(want_groups)
    assert plan.use_label and plan.use_role_text and plan.labs.shape == plan.descs.shape
    assert (plan.descs._ce_lengths == (plan.descs.argmax(dim=-1) + 1).numpy()).all()
    assert region_plan(_M(), [[None]], [desc[0][:1]], [lab[0][:1]], "desc", torch.device("cpu")) is None


def test_visible_gpus_counts_without_hip(monkeypatch):
    from clip_event_amd.launch import visible_gpus
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    n = visible_gpus()
    assert n is None or n >= 0
