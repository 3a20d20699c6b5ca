"""Image preprocessing: the numpy oracle against Pillow itself (CPU, bit for bit); the HIP kernels
against the oracle (GPU, bit for bit on the 8-bit image and on the fp32 result)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as P

SIZES = [(500, 375), (375, 500), (224, 224), (100, 80), (1024, 683), (640, 224), (231, 517), (225, 224)]


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h // 8 + 2, w // 8 + 2, 3), dtype=np.uint8)
    img = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w]          # blocky + noise: edges and flats
    noise = rng.integers(-20, 21, size=img.shape)
    return np.clip(img.astype(np.int32) + noise, 0, 255).astype(np.uint8)


def _pil_transform(img, n_px, roi=None):
    """The reference pipeline with torchvision's three transforms written out on PIL (torchvision is absent)."""
    from PIL import Image
    im = Image.fromarray(img)
    if roi is not None:
        im = im.crop(roi)
    w, h = im.size
    ow, oh = P.resized_size(w, h, n_px)
    if (ow, oh) != (w, h):
        im = im.resize((ow, oh), Image.BICUBIC)
    left, top = P.crop_offsets(ow, oh, n_px)
    im = im.crop((left, top, left + n_px, top + n_px)).convert("RGB")
    t = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    mean = torch.tensor(P.MEAN).view(3, 1, 1)
    std = torch.tensor(P.STD).view(3, 1, 1)
    return t.sub_(mean).div_(std).numpy()


@pytest.mark.parametrize("w,h", SIZES)
def test_oracle_resize_matches_pillow(w, h):
    from PIL import Image
    img = _img(w, h, w * 7 + h)
    ow, oh = P.resized_size(w, h, 224)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
    got = P.resize_bicubic_u8(img, ow, oh)
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("w,h", SIZES)
def test_oracle_transform_matches_pillow_pipeline(w, h):
    img = _img(w, h, w + 3 * h)
    assert np.array_equal(P.transform(img, 224), _pil_transform(img, 224))


def test_oracle_object_patch_matches_pillow_pipeline():
    img = _img(640, 480, 5)
    for roi in [(10, 20, 300, 200), (100, 50, 180, 400), (0, 0, 640, 480), (37, 41, 262, 266)]:
        assert np.array_equal(P.transform(img, 224, roi=roi), _pil_transform(img, 224, roi=roi)), roi


def test_crop_offsets_round_half_even():
    assert P.crop_offsets(225, 224, 224) == (0, 0) and P.crop_offsets(227, 224, 224) == (2, 0)
    assert P.resized_size(500, 375, 224) == (298, 224) and P.resized_size(224, 300, 224) == (224, 300)


@pytest.mark.gpu
def test_hip_preprocess_matches_oracle_bit_for_bit():
    """A ragged batch (8 sizes) + object boxes on two of the images, through ce_preprocess: identical fp32 bits."""
    from clip_event_amd.preprocess import preprocess
    imgs = [_img(w, h, 11 * w + h) for (w, h) in SIZES]
    rois = [None] * len(imgs)
    rois[0] = [(10, 20, 300, 200), (100, 50, 180, 360)]
    rois[4] = [(0, 0, 1024, 683), (500, 100, 900, 683)]
    out = preprocess([torch.from_numpy(a).to("cuda:0") for a in imgs], rois=rois, n_px=224)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    refs = []
    for a, r in zip(imgs, rois):
        refs.append(P.transform(a, 224))
        for box in (r or []):
            refs.append(P.transform(a, 224, roi=box))
    assert got.shape == (len(refs), 3, 224, 224)
    for i, ref in enumerate(refs):
        assert np.array_equal(got[i], ref), (i, float(np.abs(got[i] - ref).max()))


@pytest.mark.gpu
def test_hip_preprocess_rejects_bad_input():
    from clip_event_amd.preprocess import preprocess
    a = torch.zeros(50, 60, 3, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(ValueError, match="not inside"):
        preprocess([a], rois=[[(0, 0, 61, 50)]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        preprocess([a.cpu()])
    with pytest.raises(ValueError, match="uint8"):
        preprocess([a.float()])


@pytest.mark.gpu
def test_clip_transform_on_pil_images():
    """``clip._transform(n_px)`` (the ``preprocess`` returned by ``clip.load``) on PIL images: RGB and greyscale."""
    from PIL import Image
    from clip_event_amd import clip
    pre = clip._transform(224)
    rgb = _img(500, 375, 3)
    out = pre(Image.fromarray(rgb)).cpu().numpy()
    assert np.array_equal(out, _pil_transform(rgb, 224))
    grey = Image.fromarray(rgb[:, :, 0])                      # mode "L": resize-then-convert == convert-then-resize
    ref = _pil_transform(np.stack([rgb[:, :, 0]] * 3, axis=-1), 224)
    assert np.array_equal(pre(grey).cpu().numpy(), ref)
