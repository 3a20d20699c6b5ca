"""CPU: the oracle (plain-PyTorch restatement) against golden vectors captured from the
imported reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from clip_event_amd import synthetic as S
from tests.util import golden_json, golden_npz, check_summary

G = golden_json()
TINY = O.ClipConfig(**G["tiny"]["cfg"])
FP32_RTOL = 2e-4   # fp32 re-association between the reference's op order and the restatement


def test_init_params_regenerate():
    p = O.init_params(TINY, G["tiny"]["param_seed"])
    check_summary(torch.cat([v.flatten() for v in p.values()]), G["tiny"]["param_norm"], 1e-7, "tiny params")


@pytest.mark.parametrize("overbatch", [True, False])
def test_tiny_forward_backward(overbatch):
    g = G["tiny"]
    z = golden_npz("tiny_forward.npz")
    p = O.init_params(TINY, g["param_seed"])
    B, K = g["B"], g["K"]
    img = S.synthetic_images(B, TINY.image_resolution, seed=g["img_seed"])
    txt = S.synthetic_tokens(B * K, TINY.context_length, TINY.vocab_size, seed=g["txt_seed"], min_len=g["txt_min_len"])
    yi, yt, ip = O.build_labels(B, 1, K - 1, overbatch)
    ld, grads, (li, lt) = O.loss_and_grads(p, TINY, img, txt, yi, yt, ip, overbatch)
    tag = "over" if overbatch else "inst"
    np.testing.assert_allclose(li.numpy(), z[f"{tag}_logits_per_image"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lt.numpy(), z[f"{tag}_logits_per_text"], rtol=1e-4, atol=1e-4)
    assert abs(float(ld["loss_i"]) - g[tag]["loss_i"]) < 1e-5
    assert abs(float(ld["loss_t"]) - g[tag]["loss_t"]) < 1e-5
    for k, gs in g[tag]["grads"].items():
        check_summary(grads[k], gs, FP32_RTOL, k)


def test_tiny_features_and_grid():
    g = G["tiny"]
    z = golden_npz("tiny_forward.npz")
    p = O.init_params(TINY, g["param_seed"])
    img = S.synthetic_images(g["B"], TINY.image_resolution, seed=g["img_seed"])
    txt = S.synthetic_tokens(g["B"] * g["K"], TINY.context_length, TINY.vocab_size, seed=g["txt_seed"], min_len=g["txt_min_len"])
    np.testing.assert_allclose(O.encode_image(p, TINY, img).numpy(), z["image_features"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(O.encode_image(p, TINY, img, use_grid=True).numpy(), z["image_grid_features"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(O.encode_text(p, TINY, txt).numpy(), z["text_features"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kind", ["bce", "kl"])
def test_tiny_bce_kl(kind):
    g = G["tiny"]
    p = O.init_params(TINY, g["param_seed"])
    B, K = g["B"], g["K"]
    img = S.synthetic_images(B, TINY.image_resolution, seed=g["img_seed"])
    txt = S.synthetic_tokens(B * K, TINY.context_length, TINY.vocab_size, seed=g["txt_seed"], min_len=g["txt_min_len"])
    li, lt = O.clip_forward(p, TINY, img, txt, overbatch=False)
    _, yt, ip = O.build_labels(B, 1, K - 1, False)
    yb = torch.tensor([[1.] + [0.] * (K - 1)] * B)
    ld = O.criterion_contrastive(li, lt, yb, yt, ip, kind)
    assert abs(float(ld["loss_i"]) - g[kind]["loss_i"]) < 2e-5 * max(1, abs(g[kind]["loss_i"]))
    assert abs(float(ld["loss_t"]) - g[kind]["loss_t"]) < 1e-5


def test_bad_loss_kind_raises():
    with pytest.raises(RuntimeError):
        O.criterion_contrastive(torch.zeros(2, 2), torch.zeros(2, 2), None, None, torch.arange(2), "hinge")


def test_tiny_train_step():
    g = G["tiny_step"]
    p = O.init_params(TINY, 11)
    img = S.synthetic_images(4, TINY.image_resolution, seed=31)
    txt = S.synthetic_tokens(4, TINY.context_length, TINY.vocab_size, seed=32, min_len=2)
    yi, yt, ip = O.build_labels(4, 1, 0, True)
    state = {}
    for step in g["steps"]:
        p, ld, gn = O.train_step(p, TINY, state, img, txt, yi, yt, ip, lr=g["lr"], weight_decay=g["weight_decay"])
        assert abs(float(ld["loss_i"]) - step["loss_i"]) < 2e-4
        assert abs(float(ld["loss_t"]) - step["loss_t"]) < 2e-4
        assert abs(float(gn) - step["grad_norm"]) < 1e-3 * step["grad_norm"]
        for k, gs in step["params_after"].items():
            check_summary(p[k], gs, 2e-4, k)


def test_vitb32_b8_caption_only():
    """BASELINE config 1: ViT-B/32, batch 8, caption-only InfoNCE, CPU fp32."""
    g = G["vitb32"]
    z = golden_npz("vitb32_b8.npz")
    torch.set_num_threads(8)
    p = O.init_params(O.VIT_B32, g["param_seed"])
    check_summary(torch.cat([v.flatten() for v in p.values()]), g["param_norm"], 1e-7, "vit-b/32 params")
    img = S.synthetic_images(8, 224, seed=g["img_seed"])
    txt = torch.from_numpy(z["tokens"])
    y = torch.arange(8)
    ld, grads, (li, lt) = O.loss_and_grads(p, O.VIT_B32, img, txt, y, y, y, True)
    np.testing.assert_allclose(li.numpy(), z["logits_per_image"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(lt.numpy(), z["logits_per_text"], rtol=2e-4, atol=2e-4)
    assert abs(float(ld["loss_i"]) - g["loss_i"]) < 1e-4
    assert abs(float(ld["loss_t"]) - g["loss_t"]) < 1e-4
    gn, _ = O.clip_grad_norm(grads, 1.0)
    assert abs(float(gn) - g["grad_norm"]) < 1e-3 * g["grad_norm"]
    for k, gs in g["grads"].items():
        check_summary(grads[k], gs, 1e-3, k)


def test_ot_against_reference():
    z = golden_npz("ot.npz")
    g = G["ot"]
    txt = torch.from_numpy(z["txt"]).requires_grad_(True)
    obj = torch.from_numpy(z["obj"]).requires_grad_(True)
    tn, on = torch.from_numpy(z["txt_num"]), torch.from_numpy(z["obj_num"])
    tp, ip = tn == 0, on[:, 1:] == 0
    d = O.optimal_transport_dist(txt.detach(), obj.detach()[:, 1:], tp, ip)
    gd = z["dist"]
    # all-pad samples give non-finite values in the reference too: same places, same values elsewhere
    assert (np.isfinite(gd) == np.isfinite(d.numpy())).all()
    ok = np.isfinite(gd)
    np.testing.assert_allclose(d.numpy()[ok], gd[ok], rtol=1e-4, atol=1e-6)
    cost = O.cost_matrix_cosine(txt.detach(), obj.detach()[:, 1:]).masked_fill(tp.unsqueeze(-1) | ip.unsqueeze(-2), 0)
    np.testing.assert_allclose(cost.numpy(), z["cost"], rtol=1e-5, atol=1e-6)
    ld = O.criterion_alignment(txt, obj, tn, on)
    assert np.isfinite(float(ld["loss_ot"])) == g["finite"]
    if g["finite"]:
        assert abs(float(ld["loss_ot"]) - g["loss_ot"]) < 1e-5 * max(1.0, abs(g["loss_ot"]))
    # second batch: gradients
    t2 = torch.from_numpy(z["txt2"]).requires_grad_(True)
    o2 = torch.from_numpy(z["obj2"]).requires_grad_(True)
    l2 = O.criterion_alignment(t2, o2, torch.from_numpy(z["txt_num2"]), torch.from_numpy(z["obj_num2"]))["loss_ot"]
    assert abs(float(l2) - g["loss_ot2"]) < 1e-5 * max(1.0, abs(g["loss_ot2"]))
    l2.backward()
    np.testing.assert_allclose(t2.grad.numpy(), z["grad_txt2"], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(o2.grad.numpy(), z["grad_obj2"], rtol=1e-3, atol=1e-7)
    assert np.abs(o2.grad.numpy()[:, 0]).max() == 0.0      # slot 0 (whole image) never gets gradient


def test_region_branch():
    g = G["region"]
    p = O.init_params(TINY, 11)
    B = 5
    img = S.synthetic_images(B, TINY.image_resolution, seed=41)
    txt = S.synthetic_tokens(B, TINY.context_length, TINY.vocab_size, seed=42, min_len=2)
    bboxs = [[None if b is None else tuple(b) for b in bb] for bb in g["bboxs"]]
    desc = [S.synthetic_tokens(len(b), TINY.context_length, TINY.vocab_size, seed=50 + i, min_len=2) for i, b in enumerate(bboxs)]
    lab = [S.synthetic_tokens(len(b), TINY.context_length, TINY.vocab_size, seed=60 + i, min_len=2) for i, b in enumerate(bboxs)]
    for bb, pi in zip(bboxs, g["patch_idx"]):
        for b, idx in zip(bb, pi):
            if b is not None:
                assert list(O.patch_from_norm_bbox(b, TINY.grid)) == idx
    for case in g["bbox7"]:
        assert list(O.patch_from_norm_bbox(tuple(case["bbox"]), 7)) == case["idx"]
    for mode in ("desc", "desc_type", "desc_type_text"):
        q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        li, lt, lb, la = O.clip_forward_train_arg(q, TINY, img, txt, mode, bboxs, desc, lab)
        assert abs(float(lb) - g[mode]["loss_per_bbox"]) < 2e-4
        assert abs(float(la) - g[mode]["loss_per_arg"]) < 2e-4
        np.testing.assert_allclose(li.detach().numpy(), np.asarray(g[mode]["logits_per_image"]), rtol=1e-4, atol=1e-4)
        (lb + la).backward()
        for k, gs in g[mode]["grads"].items():
            if gs["norm"] == 0.0:
                assert q[k].grad is None or float(q[k].grad.norm()) == 0.0
            else:
                check_summary(q[k].grad, gs, 5e-4, f"{mode}:{k}")


def test_sim_entity_alignment():
    g = G["entity"]
    z = golden_npz("entity.npz")
    p = O.init_params(TINY, 11)
    B, O_, M = 3, 4, 5
    rng = np.random.default_rng(88)
    obj = torch.from_numpy(rng.standard_normal((B, O_, 3, TINY.image_resolution, TINY.image_resolution), dtype=np.float32))
    ent = S.synthetic_tokens(B * M, TINY.context_length, TINY.vocab_size, seed=89, min_len=2).view(B, M, -1)
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    fi, ft = O.sim_entity(q, TINY, obj, ent)
    np.testing.assert_allclose(fi.detach().numpy(), z["image_features"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(ft.detach().numpy(), z["text_features"], rtol=1e-4, atol=2e-5)
    ld = O.criterion_alignment(ft, fi, torch.from_numpy(z["ent_num"]), torch.from_numpy(z["obj_num"]))
    assert abs(float(ld["loss_ot"]) - g["loss_ot"]) < 1e-5
    ld["loss_ot"].backward()
    for k, gs in g["grads"].items():
        if gs["norm"] > 0:
            check_summary(q[k].grad, gs, 1e-3, k)


def test_label_layout_hand_derived():
    """dataset_voa.py:619,624,650-651,658,662 give worked examples in comments."""
    yi, yt, ip = O.build_labels(4, 1, 2, True)          # description_num = 3
    assert yi.tolist() == [0, 3, 6, 9]
    assert yt.tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3]
    assert ip.tolist() == [0, 3, 6, 9]
    yi, _, _ = O.build_labels(4, 1, 2, False)
    assert yi.tolist() == [0, 0, 0, 0]
    yi, yt, ip = O.build_labels(8, 1, 0, True)          # caption-only: all three = arange(8)
    assert yi.tolist() == yt.tolist() == ip.tolist() == list(range(8))


def test_lr_schedules_against_reference_golden():
    """utils.py:310-416 schedules: oracle restatement vs the lr sequence the reference's schedulers produced."""
    G = golden_json()["sched"]
    b = G["base_lr"]
    c = G["cosine"]
    got = [O.lr_warmup_cosine(b, i, c["max_iters"], warmup_epochs=c["warmup_epochs"]) for i in range(c["n"])]
    np.testing.assert_allclose(got, c["lr"], rtol=1e-12, atol=0)
    c = G["cosine_const"]
    got = [O.lr_warmup_cosine(b, i, c["max_iters"], c["warmup_factor"], c["warmup_epochs"], "constant") for i in range(c["n"])]
    np.testing.assert_allclose(got, c["lr"], rtol=1e-12, atol=0)
    c = G["multistep"]
    got = [O.lr_warmup_multistep(b, i, c["milestones"], c["gamma"], warmup_epochs=c["warmup_epochs"]) for i in range(c["n"])]
    np.testing.assert_allclose(got, c["lr"], rtol=1e-12, atol=0)
