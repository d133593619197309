"""GPU parity for the 8-bit image path through the C ABI: bit-exact against the integer oracle
(resize, BGR<->Lab + CLAHE), 5e-4 on the normalised network input, 1e-3 on crop logits."""
import numpy as np
import pytest
import torch

from oracle import b0_ref
from oracle import imgproc_ref as R

pytestmark = pytest.mark.gpu


def _frame(h, w, seed):
    rs = np.random.RandomState(seed)
    base = rs.randint(50, 200, (h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    ell = ((xx - w // 2) / (w / 6.0)) ** 2 + ((yy - h // 2) / (h / 4.0)) ** 2 <= 1.0     # a flat "face"
    base[ell] = (180, 160, 140)
    return base


@pytest.mark.parametrize("h,w,dh,dw", [(1080, 1920, 256, 256), (480, 640, 300, 300), (120, 160, 256, 256),
                                         (300, 500, 300, 300), (256, 256, 256, 256), (31, 47, 64, 80)])
def test_resize_bit_exact(b0_handle, h, w, dh, dw):
    f = _frame(h, w, h + w)
    got = b0_handle.resize_bgr(f, dw, dh)
    want = R.resize_linear_u8(f, dw, dh)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{(got != want).sum()} differing bytes"


def test_resize_strided_view(b0_handle):
    f = _frame(200, 300, 9)
    view = f[10:150, 20:260]                       # row stride != width*3
    assert np.array_equal(b0_handle.resize_bgr(np.ascontiguousarray(view), 64, 64),
                          R.resize_linear_u8(np.ascontiguousarray(view), 64, 64))


@pytest.mark.parametrize("h,w", [(224, 224), (160, 128), (83, 101), (37, 64), (480, 640)])
def test_preprocess_face_quality_bit_exact(b0_handle, h, w):
    f = _frame(h, w, 3 * h + w)
    got = b0_handle.preprocess_face_quality(f)
    want = R.preprocess_face_quality(f)
    assert np.array_equal(got, want), f"{(got != want).sum()} of {got.size} bytes differ; max {np.abs(got.astype(int) - want).max()}"


def test_preprocess_crops_matches_oracle(b0_handle):
    f = _frame(1080, 1920, 7)
    boxes = [(800, 300, 320, 440), (10, 20, 64, 64), (1700, 900, 220, 180), (0, 0, 1920, 1080), (500, 500, 79, 233)]
    for clahe in (False, True):
        got = b0_handle.preprocess_crops(f, boxes, apply_clahe=clahe)
        for i, (x, y, w, h) in enumerate(boxes):
            face = f[y:y + h, x:x + w]
            if clahe:
                face = R.preprocess_face_quality(face)
            want = R.crop_resize_normalize(face)
            err = np.abs(got[i] - want).max()
            # float32 source coordinates near x=1900 carry ~1e-4 of absolute rounding, which the
            # blend weight inherits; values are u8/255/std, so 5e-4 is a few ulps of that weight
            assert err <= 5e-4, (clahe, i, err)


def test_classify_crops_matches_oracle(pkg, b0_handle, seeded_sd):
    f = _frame(720, 1280, 11)
    boxes = [(500, 200, 300, 380), (30, 40, 120, 90), (900, 400, 224, 224)]
    got = b0_handle.classify_crops(f, boxes, apply_clahe=True)
    x = np.stack([R.crop_resize_normalize(R.preprocess_face_quality(f[y:y + h, x:x + w])) for (x, y, w, h) in boxes])
    want = b0_ref.forward(pkg.weights.to_torch(seeded_sd), torch.from_numpy(x)).numpy()
    assert np.abs(got - want).max() <= 1e-3, (got.ravel(), want.ravel())


def test_bad_boxes_fail_loudly(pkg, b0_handle):
    f = _frame(100, 100, 1)
    with pytest.raises(pkg._lib.DfdError):
        b0_handle.preprocess_crops(f, [(90, 90, 20, 20)])           # leaves the frame
    with pytest.raises(pkg._lib.DfdError):
        b0_handle.preprocess_crops(f, [(0, 0, 0, 10)])
