"""BASELINE.json's full sizes (ViT-B/32, per-GPU batch 256, 77-token captions) checked through size-independent
properties -- the CPU oracle cannot run them in seconds.  The small-size oracle / golden comparisons are in
test_model_gpu.py; here the SAME code path is exercised at the benchmarked shape."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def vitb32():
    from oracle import clip_oracle as O
    from clip_event_amd.model import build_model
    return build_model(O.init_params(O.VIT_B32, 0)).to(DEV)


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_batch_permutation_equivariance_and_logit_symmetry(vitb32):
    """Every sample is processed independently until the logits, and a row's arithmetic does not depend on where it
    sits in a tile: features of a permuted batch are the permuted features BIT FOR BIT (images: dense rows; captions:
    packed rows move to different offsets and lengths).  K = 1 over-batch logits are each other's transposes."""
    from clip_event_amd import synthetic as S
    m = vitb32
    B = 256
    img = S.synthetic_images(B, 224, seed=999).to(DEV)
    txt = S.synthetic_tokens(B, 77, 49408, seed=999).to(DEV)
    perm = torch.from_numpy(np.random.default_rng(0).permutation(B)).to(DEV)
    with torch.no_grad():
        fi, ft = m.encode_both(img, txt)
        fi_p, ft_p = m.encode_both(img[perm].contiguous(), txt[perm].contiguous())
        lpi, lpt = m(img, txt)
    torch.cuda.synchronize()
    assert torch.equal(fi_p, fi[perm]), "image features are not permutation-equivariant bit for bit"
    assert torch.equal(ft_p, ft[perm]), "text features are not permutation-equivariant bit for bit"
    assert tuple(lpi.shape) == (B, B) and bool(torch.isfinite(lpi).all())
    assert float((lpi - lpt.t()).abs().max()) < 1e-3 * float(lpi.abs().max())
    # unit-norm rows behind the logits: |logit| <= exp(logit_scale)
    assert float(lpi.abs().max()) <= float(m.logit_scale.detach().exp()) * (1 + 1e-5)


def test_full_size_step_packed_equals_dense(vitb32):
    """One full training-shaped forward + backward at B = 256 with the text tower on live tokens vs on all 77
    positions: same loss, same gradients (tile-order rounding only), and the packed run's caption rows after the
    EOT really are absent (rows < B*77)."""
    from clip_event_amd import synthetic as S
    from clip_event_amd.functional import text_packing
    from clip_event_amd.losses import CriterionContrastive
    m = vitb32
    B = 256
    img = S.synthetic_images(B, 224, seed=7).to(DEV)
    txt = S.synthetic_tokens(B, 77, 49408, seed=7).to(DEV)
    y = torch.arange(B, device=DEV)
    crit = CriterionContrastive("ce")
    out = {}
    for packed in (True, False):
        m.pack_text = packed
        m.zero_grad(set_to_none=True)
        ld = crit(*m(img, txt), y, y, index_pos=y)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        out[packed] = (float(ld["loss_i"]), float(ld["loss_t"]), m._flat_grad.detach().clone())
    m.pack_text = True
    pk = text_packing(m, txt)
    assert pk.rows == int((txt.argmax(dim=-1) + 1).sum()) and pk.rows < B * 77
    (li_p, lt_p, g_p), (li_d, lt_d, g_d) = out[True], out[False]
    assert abs(li_p - li_d) < 1e-4 and abs(lt_p - lt_d) < 1e-4
    assert bool(torch.isfinite(g_p).all()) and float(g_p.norm()) > 0
    assert _rel(g_p, g_d) < 2e-3, _rel(g_p, g_d)
    # per tower as well (a tower-level slip would hide in the total)
    for name, (a, b) in m._ranges.items():
        if b > a and float(g_d[a:b].norm()) > 0:
            assert _rel(g_p[a:b], g_d[a:b]) < 3e-3, (name, _rel(g_p[a:b], g_d[a:b]))


def test_full_size_step_is_reproducible(vitb32):
    """Two streams, side-stream gradient writes, lazy and eager gradient zeroing: repeated identical steps must agree
    up to the order of the fp32 atomic adds (1e-6), whichever zero_grad flavour precedes them.  A cross-stream race
    shows up here as a per-step difference of several per cent."""
    from clip_event_amd import synthetic as S
    from clip_event_amd.losses import CriterionContrastive
    m = vitb32
    B = 256
    img = S.synthetic_images(B, 224, seed=7).to(DEV)
    txt = S.synthetic_tokens(B, 77, 49408, seed=7).to(DEV)
    y = torch.arange(B, device=DEV)
    crit = CriterionContrastive("ce")

    def run(set_none):
        m.zero_grad(set_to_none=set_none)
        ld = crit(*m(img, txt), y, y, index_pos=y)
        sum(ld.values()).backward()
        torch.cuda.synchronize()
        return m._flat_grad.detach().clone()

    ref = run(True)
    for it in range(6):
        assert _rel(run(it % 2 == 0), ref) < 1e-6


def test_full_size_hard_negative_rows(vitb32):
    """Config c3's shape on one GPU at reduced batch (B = 64 images x K = 5 descriptions = 320 captions): the logits of
    the full [B, B*K] problem restricted to a sub-batch equal the sub-batch run alone (rows are independent), and the
    positive-row logits_per_text block matches index_pos."""
    from clip_event_amd import synthetic as S
    from clip_event_amd import distributed as D
    m = vitb32
    B, K = 64, 5
    img = S.synthetic_images(B, 224, seed=3).to(DEV)
    txt = S.synthetic_tokens(B * K, 77, 49408, seed=4).to(DEV)
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=DEV, rank_=0)
    assert yi.tolist()[:3] == [0, 5, 10] and ip.tolist()[:3] == [0, 5, 10] and yt.tolist()[:6] == [0, 0, 0, 0, 0, 1]
    with torch.no_grad():
        lpi, lpt = m(img, txt)
        sub = slice(0, 16)
        lpi_s, lpt_s = m(img[sub].contiguous(), txt[:16 * K].contiguous())
    assert tuple(lpi.shape) == (B, B * K) and tuple(lpt.shape) == (B * K, B)
    assert float((lpi[sub, :16 * K] - lpi_s).abs().max()) < 2e-3
    assert float((lpt[:16 * K, sub] - lpt_s).abs().max()) < 2e-3


def test_fixed_batch_is_fitted():
    """Training dynamics end to end (fused clip + Adam, warm-up schedule, packed text, two streams): 30 steps on one
    fixed batch of 32 pairs drive the InfoNCE loss from ~ln(32)*2 to below 5 % of it."""
    from oracle import clip_oracle as O
    from clip_event_amd import synthetic as S, distributed as D
    from clip_event_amd.engine import train_step
    from clip_event_amd.losses import CriterionContrastive
    from clip_event_amd.model import build_model
    from clip_event_amd.optim import FusedAdam, WarmupCosineLR
    m = build_model(O.init_params(O.VIT_B32, 1)).to(DEV)
    B = 32
    opt = FusedAdam(m, lr=2e-5, max_norm=1.0)
    sch = WarmupCosineLR(opt, max_iters=60, warmup_epochs=5)
    img = S.synthetic_images(B, 224, seed=1).to(DEV)
    txt = S.synthetic_tokens(B, 77, 49408, seed=2).to(DEV)
    yi, yt, ip = D.global_labels(B, 1, 0, True, device=DEV, rank_=0)
    crit = CriterionContrastive("ce")
    losses = []
    for _ in range(30):
        ld = train_step(m, crit, opt, img, txt, yi, yt, ip)
        sch.step()
        losses.append(float(sum(v.detach() for v in ld.values())))
    assert all(l == l for l in losses)
    print("loss", [round(l, 3) for l in losses[::5]], "->", losses[-1])
    assert losses[-1] < 0.05 * losses[0]


def test_fused_head_equals_logits_plus_criterion_path():
    """ViT-B/32, B = 8, K = 3 hard negatives: the fused head (no logits matrix, engine default) and the reference-shaped
    path (model.forward logits + CriterionContrastive) give the same losses (1e-5) and the same parameter gradients to
    bf16 rounding (relative 5e-3, measured 2.1e-3: the two fp32 heads agree to 1e-7, but the towers' backward rounds
    the feature gradient to bf16 first, so 1-ulp differences flip roundings -- the level of the packed-vs-dense test)."""
    import os
    from clip_event_amd import synthetic as S, distributed as D
    from clip_event_amd.engine import contrastive_step_losses
    from clip_event_amd.losses import CriterionContrastive
    m = S.synthetic_model("vit_b32", seed=3).to(DEV)
    B, K = 8, 3
    img = S.synthetic_images(B, 224, seed=5).to(DEV)
    txt = S.synthetic_tokens(B * K, 77, 49408, seed=6).to(DEV)
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=DEV, rank_=0)
    crit = CriterionContrastive("ce")
    out = {}
    for fused in ("1", "0"):
        os.environ["CE_FUSED_HEAD"] = fused
        try:
            m.zero_grad()
            ld = contrastive_step_losses(m, crit, img, txt, yi, yt, ip)
            sum(ld.values()).backward()
            torch.cuda.synchronize()
            out[fused] = ({k: float(v) for k, v in ld.items()}, m._flat_grad.clone())
        finally:
            os.environ.pop("CE_FUSED_HEAD", None)
    print("fused", out["1"][0], "unfused", out["0"][0])
    for k in ("loss_i", "loss_t"):
        assert abs(out["1"][0][k] - out["0"][0][k]) < 1e-5 * max(1.0, abs(out["0"][0][k]))
    rel = float((out["1"][1] - out["0"][1]).norm() / out["0"][1].norm())
    print("flat gradient rel-l2 fused vs unfused:", rel)
    assert rel < 5e-3


def test_small_head_equals_logits_plus_criterion_path():
    """ViT-B/32, B = 16, K = 1 and K = 3: the three-launch head engine.train_step takes below 2^17 logits (csrc/head_small.hip)
    and the reference-shaped path (logits + CriterionContrastive; CE_SMALL_HEAD=0) give the same losses (1e-5) and the same
    parameter gradients to bf16 rounding (relative 5e-3: the two fp32 heads agree to 1e-6, the towers' backward rounds the
    feature gradient to bf16 first)."""
    import os
    from clip_event_amd import synthetic as S, distributed as D
    from clip_event_amd.engine import contrastive_step_losses
    from clip_event_amd.losses import CriterionContrastive
    m = S.synthetic_model("vit_b32", seed=3).to(DEV)
    crit = CriterionContrastive("ce")
    for B, K in ((16, 1), (8, 3)):
        img = S.synthetic_images(B, 224, seed=5).to(DEV)
        txt = S.synthetic_tokens(B * K, 77, 49408, seed=6).to(DEV)
        yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=DEV, rank_=0)
        out = {}
        for small in ("1", "0"):
            os.environ["CE_SMALL_HEAD"] = small
            try:
                m.zero_grad()
                ld = contrastive_step_losses(m, crit, img, txt, yi, yt, ip)
                sum(ld.values()).backward()
                torch.cuda.synchronize()
                out[small] = ({k: float(v) for k, v in ld.items()}, m._flat_grad.clone())
            finally:
                os.environ.pop("CE_SMALL_HEAD", None)
        print("small head", out["1"][0], "general head", out["0"][0])
        for k in ("loss_i", "loss_t"):
            assert abs(out["1"][0][k] - out["0"][0][k]) < 1e-5 * max(1.0, abs(out["0"][0][k]))
        rel = float((out["1"][1] - out["0"][1]).norm() / out["0"][1].norm())
        print("flat gradient rel-l2 small vs general head:", rel)
        assert rel < 5e-3


def test_full_size_fp16_stream_step(vitb32):
    """The benchmarked shape (B = 256) with the residual / gradient stream in IEEE fp16 (model.stream16) against the fp32
    stream on the SAME weights and batch: same loss (1e-3), every tower's gradient within the bf16-operand noise floor of
    the fp32-stream build (relative L2 <= 3e-2; the fp16 stream adds 2^-11 relative rounding per write, the bf16 GEMM
    operands 2^-8), nothing saturated (a stream value at +-65504 would turn up as a gradient error of order 1), and the
    fp16 run repeats bit-for-bit in its forward (features)."""
    from clip_event_amd import synthetic as S
    from clip_event_amd.losses import CriterionContrastive
    m = vitb32
    B = 256
    img = S.synthetic_images(B, 224, seed=11).to(DEV)
    txt = S.synthetic_tokens(B, 77, 49408, seed=12).to(DEV)
    y = torch.arange(B, device=DEV)
    crit = CriterionContrastive("ce")
    out = {}
    keep = m.stream16
    try:
        for s16 in (False, True):
            m.stream16 = s16
            m.zero_grad(set_to_none=True)
            ld = crit(*m(img, txt), y, y, index_pos=y)
            sum(ld.values()).backward()
            torch.cuda.synchronize()
            with torch.no_grad():
                fi, ft = m.encode_both(img, txt)
            out[s16] = (float(ld["loss_i"].detach()), float(ld["loss_t"].detach()), m._flat_grad.detach().clone(), fi.clone(), ft.clone())
        with torch.no_grad():
            fi2, ft2 = m.encode_both(img, txt)        # still stream16
        assert torch.equal(fi2, out[True][3]) and torch.equal(ft2, out[True][4])
    finally:
        m.stream16 = keep
    (li32, lt32, g32, fi32, ft32), (li16, lt16, g16, fi16, ft16) = out[False], out[True]
    print(f"loss_i {li16:.5f} / {li32:.5f}  loss_t {lt16:.5f} / {lt32:.5f}  features rel {_rel(fi16, fi32):.2e} / {_rel(ft16, ft32):.2e}  "
          f"flat gradient rel {_rel(g16, g32):.3e}")
    assert abs(li16 - li32) < 1e-3 and abs(lt16 - lt32) < 1e-3
    # two builds that each sit 3-5e-3 from the same-rounding oracle (bf16 roundings that flip with any upstream change):
    # 1e-2 between them.  Measured (round 3): image 3.0e-3, text 5.5e-3 -- a drift
    # past 7e-3 is reported as a warning long before the bound trips.  This is a self-comparison; the checks that face the
    # REFERENCE run the fp16 stream at the fp32 stream's tolerances: test_vitb32_b8_against_reference_golden[stream16] and
    # test_vitb32_b8_gradient_error_is_at_the_bf16_noise_floor[stream16] (tests/test_model_gpu.py), and, at trained-like
    # statistics, tests/test_stream16_hostile_gpu.py.
    assert _rel(fi16, fi32) < 1e-2 and _rel(ft16, ft32) < 1e-2
    if max(_rel(fi16, fi32), _rel(ft16, ft32)) > 7e-3:
        import warnings
        warnings.warn(f"fp16-stream vs fp32-stream features drifted: {_rel(fi16, fi32):.2e} / {_rel(ft16, ft32):.2e} (measured 3.0e-3 / 5.5e-3)")
    assert bool(torch.isfinite(g16).all())
    for name, (a, b) in m._ranges.items():
        if b > a and float(g32[a:b].norm()) > 0:
            r = _rel(g16[a:b], g32[a:b])
            print(f"  {name}: gradient rel-L2 fp16 stream vs fp32 stream {r:.3e}")
            assert r < 3e-2, (name, r)


def test_config3_per_rank_full_size(vitb32):
    """BASELINE config 3 at its PER-RANK size: B = 512 images x K = 5 descriptions (1 positive + 2 event + 2 argument
    negatives, dataset_voa.py:605-664) = 2,560 captions, labels of rank 0 -- the largest single-GPU configuration.  The CPU
    oracle cannot run it, so size-independent properties: (1) label layout of `global_labels` (bit-exact integers); (2) one
    full `engine.train_step`-shaped forward + backward through the fused head is finite and non-trivial; (3) the loss is
    invariant (1e-4) under permuting the images TOGETHER with their description groups -- every row is independent until
    the head and the head is a sum over rows; (4) the image / text features of the first 32 images and their 160 captions
    equal the features of that sub-batch run alone (2e-3: other tile heights at most; packed captions move to other offsets)."""
    from clip_event_amd import synthetic as S
    from clip_event_amd import distributed as D
    from clip_event_amd.engine import contrastive_step_losses
    from clip_event_amd.losses import CriterionContrastive
    m = vitb32
    B, K = 512, 5
    img = S.synthetic_images(B, 224, seed=21).to(DEV)
    txt = S.synthetic_tokens(B * K, 77, 49408, seed=22).to(DEV)
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, device=DEV, rank_=0)
    assert yi.tolist()[:3] == [0, 5, 10] and yi.numel() == B and int(yi[-1]) == (B - 1) * K
    assert yt.tolist()[:6] == [0, 0, 0, 0, 0, 1] and yt.numel() == B * K
    assert ip.tolist()[:3] == [0, 5, 10] and ip.numel() == B
    crit = CriterionContrastive("ce")
    m.zero_grad(set_to_none=True)
    ld = contrastive_step_losses(m, crit, img, txt, yi, yt, ip)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    li, lt = float(ld["loss_i"].detach()), float(ld["loss_t"].detach())
    g = m._flat_grad.detach().clone()
    print(f"config 3 per rank: loss_i {li:.4f} (ln {B * K} = {np.log(B * K):.4f}), loss_t {lt:.4f} (ln {B} = {np.log(B):.4f}), |g| {float(g.norm()):.4f}")
    assert bool(torch.isfinite(g).all()) and float(g.norm()) > 0
    assert abs(li - np.log(B * K)) < 1.0 and abs(lt - np.log(B)) < 1.0          # random init: near-uniform softmax
    perm = torch.from_numpy(np.random.default_rng(5).permutation(B)).to(DEV)
    tperm = (perm[:, None] * K + torch.arange(K, device=DEV)[None, :]).reshape(-1)
    with torch.no_grad():
        ldp = contrastive_step_losses(m, crit, img[perm].contiguous(), txt[tperm].contiguous(), yi, yt, ip)
        fi, ft = m.encode_both(img, txt)
        fi_s, ft_s = m.encode_both(img[:32].contiguous(), txt[:32 * K].contiguous())
    assert abs(float(ldp["loss_i"]) - li) < 1e-4 * max(1.0, li) and abs(float(ldp["loss_t"]) - lt) < 1e-4 * max(1.0, lt)
    # rows are independent; the sub-batch's GEMMs may pick other tile heights (same K order) -- bf16 rounding level at most
    print(f"sub-batch features rel {_rel(fi[:32], fi_s):.2e} / {_rel(ft[:32 * K], ft_s):.2e}")
    assert _rel(fi[:32], fi_s) < 2e-3 and _rel(ft[:32 * K], ft_s) < 2e-3


def test_full_size_adam_in_tiles_equals_the_flat_kernel(monkeypatch):
    """ViT-B/32's real parameter set (151 M parameters, block weights 2304 x 768 / 768 x 768 / 3072 x 768 / 768 x 3072 and the
    512-wide text tower): `ce_adam_step_tiles` against the flat kernel -- masters, both moments and the bf16 mirror bit for bit on
    every parameter element, every W^T operand copy the transpose of its mirror, after two steps with weight decay (max_norm far
    above the norm: the clip coefficient is exactly 1 whatever order the norm's float atomics arrived in)."""
    from oracle import clip_oracle as O
    from clip_event_amd.model import build_model
    from clip_event_amd.optim import FusedAdam
    out = {}
    for tiles in ("1", "0"):
        monkeypatch.setenv("CE_ADAM_TILES", tiles)
        m = build_model(O.init_params(O.VIT_B32, 0)).to(DEV)
        opt = FusedAdam(m, lr=1e-4, max_norm=1e9, weight_decay=0.01)
        opt.zero_grad()
        g = torch.Generator(device=DEV).manual_seed(3)
        for it in range(2):
            m._flat_grad.copy_(torch.randn(m._flat_grad.numel(), generator=g, device=DEV) * 0.02)
            opt.step()
        torch.cuda.synchronize()
        if tiles == "1":
            assert m._wt_fresh
            for n in m._pmap:
                if m._is_block_weight(n):
                    assert torch.equal(m._w16t[n], m._w16[n].t().contiguous()), n
        live = torch.zeros(m._flat.numel(), dtype=torch.bool, device=DEV)
        for n, p_ in m._pmap.items():
            live[m._offsets[n]: m._offsets[n] + p_.numel()] = True
        out[tiles] = tuple(t[live].clone() for t in (m._flat, opt.m, opt.v, m._flat16))
        del m, opt
        torch.cuda.empty_cache()
    for i, what in enumerate(("masters", "exp_avg", "exp_avg_sq", "bf16 mirror")):
        assert torch.equal(out["1"][i], out["0"][i]), what
