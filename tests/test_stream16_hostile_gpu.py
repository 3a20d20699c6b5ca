"""The fp16 residual / gradient stream at TRAINED-LIKE statistics, and its saturation telemetry.

Every other fixture is random-init, where activations are O(1).  Here the same comparisons against the CPU oracle (fp32 =
the reference's arithmetic, model_clip.py:190-200 keeps the stream in fp32) run on a state with outlier residual channels of
magnitude 100-300, LayerNorm gains up to x10, ``logit_scale = ln 100`` and captions of 3 and of context-length tokens
(tests/hostile.py) -- in BOTH stream formats with the SAME bounds, and with the device-side clamp counters asserted zero.
A second group forces the clamp (an activation above 65504; a gradient scale with no head-room) and checks that it is
counted, raised by the asynchronous poll and stops ``train_step(check_finite=True)``.

Bounds: those of tests/test_model_gpu.py (worst gradient cosine > 0.98 -- 0.97 on ViT-B/32, where the oracle's own
bf16-operand mode reads 0.984; and the fp16 stream held to the fp32 stream per parameter --, median relative L2 < 0.03, |d loss| < 2e-2), with
the two that are absolute on the logits scaled to this state's logit scale (100 instead of 14.3: logits within 0.15 * 7 of
the fp32 oracle; losses within 2e-2 * max(1, |loss|) -- loss_t is ~10 here).  What the oracle's own bf16-operand mode
measures on this state (CPU, /tmp-free: tools/diag/hostile_calibration.py): tiny logits 0.11-0.13, loss_t 0.055-0.060,
worst cosine 0.9995, median rel-L2 0.008, the same with and without the fp16 stream rounding points."""
import contextlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
LOGIT_TOL = 0.15 * 100.0 / math.exp(math.log(1 / 0.07))      # the suite's 0.15 at scale 14.29, carried to scale 100


def _cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _geometry(name):
    from oracle import clip_oracle as O
    if name == "tiny":
        return O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 2), 6, 11
    return O.VIT_B32, 8, 0


@pytest.mark.parametrize("geometry", ["tiny", "vit_b32_b8"])
def test_trained_like_statistics_against_oracle(geometry, monkeypatch):
    """Both stream formats on one trained-like state, each against the fp32 oracle with the suite's bounds, then against each
    other: the fp16 stream may not be worse than the fp32 stream (whole-gradient error <= 1.25x + 0.003, per-parameter
    error distribution: median <= 1.3x, 90th percentile <= 1.5x) -- if it were, the default would have to go back to fp32
    (VERDICT r3 item 4)."""
    from oracle import clip_oracle as O
    from clip_event_amd import synthetic as S
    from clip_event_amd.losses import CriterionContrastive
    from clip_event_amd.model import build_model
    from tests.hostile import hostile_state, hostile_tokens
    cfg, B, seed = _geometry(geometry)
    sd = hostile_state(O.init_params(cfg, seed), cfg, seed=seed)
    img = S.synthetic_images(B, cfg.image_resolution, seed=31)
    txt = hostile_tokens(B, cfg.context_length, cfg.vocab_size, seed=32)
    lens = (txt.argmax(-1) + 1).tolist()
    assert lens[0] == 3 and lens[1] == cfg.context_length
    y = torch.arange(B)
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    ld32, g32, (li32, lt32) = O.loss_and_grads(sd, cfg, img, txt, y, y, y, True)
    res = {}
    for stream16 in (False, True):
        monkeypatch.setenv("CE_STREAM16", "1" if stream16 else "0")
        with (O.stream_f16() if stream16 else contextlib.nullcontext()):
            _, g16, (li16, lt16) = O.loss_and_grads(sd, cfg, img, txt, y, y, y, True, bf16=True)   # the build's rounding points
        m = build_model({k: v.clone() for k, v in sd.items()}).to(DEV)
        assert m.stream16 == stream16
        m._ready()
        m.stream16_saturation(reset=True)
        li, lt = m(img.to(DEV), txt.to(DEV))
        ld = CriterionContrastive("ce")(li, lt, y.to(DEV), y.to(DEV), index_pos=y.to(DEV))
        (ld["loss_i"] + ld["loss_t"]).backward()
        torch.cuda.synchronize()
        sat = m.stream16_saturation()
        d32 = max(float((li.cpu() - li32).abs().max()), float((lt.cpu() - lt32).abs().max()))
        d16 = max(float((li.cpu() - li16).abs().max()), float((lt.cpu() - lt16).abs().max()))
        worst, worst_ln, rel, floor = (1.0, None), (1.0, None), {}, {}
        for n, p in m.named_parameters():
            g = g32[n]
            if g is None or float(g.norm()) == 0.0:
                continue
            c = _cos(p.grad, g)
            rel[n], floor[n] = _rel(p.grad, g), _rel(g16[n], g)
            if _is_ln(n):
                if c < worst_ln[0]:
                    worst_ln = (c, n)
            elif c < worst[0]:
                worst = (c, n)
        names = [n for n, _ in m.named_parameters() if g32[n] is not None]
        flat = torch.cat([dict(m.named_parameters())[n].grad.flatten().double().cpu() for n in names])
        flat32 = torch.cat([g32[n].flatten().double() for n in names])
        flat16 = torch.cat([g16[n].flatten().double() for n in names])
        whole, whole_floor = float((flat - flat32).norm() / flat32.norm()), float((flat16 - flat32).norm() / flat32.norm())
        norm_ratio = float(flat.norm() / flat32.norm())
        rels = np.array(list(rel.values()))
        print(f"[{geometry} stream16={stream16}] logits vs fp32 {d32:.3f} (range {float(li32.abs().max()):.1f}), vs same-rounding oracle {d16:.3f}; "
              f"loss_i {float(ld['loss_i']):.4f}/{float(ld32['loss_i']):.4f} loss_t {float(ld['loss_t']):.4f}/{float(ld32['loss_t']):.4f}; "
              f"WHOLE gradient rel-l2 {whole:.4f} (bf16-operand oracle {whole_floor:.4f}); per parameter: median {np.median(rels):.4f} "
              f"p90 {np.quantile(rels, 0.9):.4f} max {rels.max():.4f} (oracle: median {np.median(list(floor.values())):.4f} max {max(floor.values()):.4f}); "
              f"worst cosine {worst[0]:.5f} at {worst[1]}, among LayerNorm parameters {worst_ln[0]:.5f} at {worst_ln[1]}; "
              f"total norm ratio {norm_ratio:.4f}; clamp counters {sat}")
        assert d32 < LOGIT_TOL and d16 < LOGIT_TOL / 3
        for k in ("loss_i", "loss_t"):
            assert abs(float(ld[k]) - float(ld32[k])) < 2e-2 * max(1.0, abs(float(ld32[k])))
        # The suite's bounds (worst cosine 0.98, median relative error 0.03) for every parameter but the LayerNorm gains / biases:
        # behind outlier channels their gradient is two or three entries (the outlier channels': 0.0020 and 0.0018 of a 0.0033
        # norm in visual block 11's ln_1) that are themselves small differences of large sums, and bf16 operand rounding moves
        # them at random -- the ORACLE's bf16-operand mode against its own fp32 mode reads 0.17-0.31 worst relative error on
        # this state, on a different LayerNorm for every image seed and with no preference for either stream format
        # (tools/diag/hostile_sensitivity.py).  They get a looser bound (0.8; 0.98 on the tiny geometry, where it holds), the same
        # for both formats, and the comparison that discriminates is statistical: the whole gradient vector, and the
        # distribution of the per-parameter errors, fp16 stream against fp32 stream (below).
        assert worst[0] > 0.98 and np.median(rels) < 0.03
        assert worst_ln[0] > (0.98 if geometry == "tiny" else 0.8)
        assert whole < 0.03 and abs(norm_ratio - 1.0) < 0.02
        assert sat == (0, 0), f"fp16 stream clamped at trained-like statistics: {sat}"
        res[stream16] = (whole, rels)
        del m
    (w32, r32), (w16, r16) = res[False], res[True]
    print(f"[{geometry}] fp16 vs fp32 stream: whole-gradient error {w16:.4f} vs {w32:.4f}; per-parameter median {np.median(r16):.4f} vs "
          f"{np.median(r32):.4f}, p90 {np.quantile(r16, 0.9):.4f} vs {np.quantile(r32, 0.9):.4f}")
    assert w16 < 1.25 * w32 + 0.003
    assert np.median(r16) < 1.3 * np.median(r32) + 0.002 and np.quantile(r16, 0.9) < 1.5 * np.quantile(r32, 0.9) + 0.01


def _is_ln(name):
    return ".ln_" in name or name.startswith(("ln_final", "visual.ln_"))


def _tiny_model(monkeypatch, s16=True):
    from oracle import clip_oracle as O
    from clip_event_amd.model import build_model
    monkeypatch.setenv("CE_STREAM16", "1" if s16 else "0")
    cfg = O.ClipConfig(64, 64, 2, 128, 32, 20, 512, 128, 2, 3)
    sd = O.init_params(cfg, 5)
    return cfg, sd, build_model


def _step(m, cfg, B=6):
    from clip_event_amd import synthetic as S
    from clip_event_amd.losses import CriterionContrastive
    img = S.synthetic_images(B, cfg.image_resolution, seed=3).to(DEV)
    txt = S.synthetic_tokens(B, cfg.context_length, cfg.vocab_size, seed=4, min_len=2).to(DEV)
    y = torch.arange(B, device=DEV)
    li, lt = m(img, txt)
    ld = CriterionContrastive("ce")(li, lt, y, y, index_pos=y)
    (ld["loss_i"] + ld["loss_t"]).backward()
    torch.cuda.synchronize()
    return img, txt, y


def test_forward_stream_clamp_is_counted(monkeypatch):
    """An activation beyond 65504 (a c_proj bias of 1e5 on one channel): the fp16 stream clamps it, the forward counter
    says so; the fp32 stream neither clamps nor counts."""
    for s16 in (True, False):
        cfg, sd, build_model = _tiny_model(monkeypatch, s16)
        sd = {k: v.clone() for k, v in sd.items()}
        sd["visual.transformer.resblocks.0.mlp.c_proj.bias"][7] = 1.0e5
        m = build_model(sd).to(DEV)
        m._ready()
        m.stream16_saturation(reset=True)
        _step(m, cfg)
        f, g = m.stream16_saturation(reset=True)
        print(f"stream16={s16}: counters forward {f} gradient {g}")
        if s16:
            assert f > 0
        else:
            assert (f, g) == (0, 0)


def test_gradient_stream_clamp_is_counted_polled_and_stops_training(monkeypatch):
    """No head-room in the gradient scale (grad_target just under the fp16 limit): whatever the gradient grows by on its way
    down the tower is clipped.  The counter sees it, the asynchronous poll raises ``Stream16Saturation`` (second call: it
    examines the copy the first one started), and ``train_step(check_finite=True)`` stops the run as the reference stops on a
    non-finite loss (engine.py:79-82)."""
    from clip_event_amd.engine import train_step
    from clip_event_amd.losses import CriterionContrastive
    from clip_event_amd.model import Stream16Saturation
    from clip_event_amd.optim import FusedAdam
    cfg, sd, build_model = _tiny_model(monkeypatch, True)
    m = build_model({k: v.clone() for k, v in sd.items()}).to(DEV)
    m._ready()
    m.grad_target = 65504.0 * 0.9
    m.stream16_saturation(reset=True)
    img, txt, y = _step(m, cfg)
    f, g = m.stream16_saturation()
    print(f"counters forward {f} gradient {g}")
    assert f == 0 and g > 0
    m._sat_poll = None
    m.poll_stream16_saturation()            # starts the copy
    torch.cuda.synchronize()
    with pytest.raises(Stream16Saturation):
        m.poll_stream16_saturation()        # examines it
    opt = FusedAdam(m, lr=1e-6)
    with pytest.raises(SystemExit):
        train_step(m, CriterionContrastive("ce"), opt, img, txt, y, y, y, check_finite=True)
    # with head-room restored and the counters cleared the same step runs clean
    m.grad_target = 64.0
    m.stream16_saturation(reset=True)
    train_step(m, CriterionContrastive("ce"), opt, img, txt, y, y, y, check_finite=True)
    assert m.stream16_saturation() == (0, 0)
