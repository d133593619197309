"""The OpenCV restatements in oracle/imgproc_ref.py against independent implementations that ARE installed here
(cv2 is not): scipy.ndimage for the linear filters, the CIE / colorsys definitions in float64 for the fixed-point colour
conversions.  These do not pin the oracle to cv2's bits (nothing here can), they bound how far a restatement could be
from the operation it names: exact for the integer filters, within one or two 8-bit levels for the colour spaces."""
import colorsys

import numpy as np
import scipy.ndimage as ndi

from oracle import imgproc_ref as R


def _gray(seed, h=96, w=128):
    return np.random.RandomState(seed).randint(0, 256, (h, w)).astype(np.uint8)


def test_laplacian_and_sobel_equal_scipy_mirror_filters():
    g = _gray(1)
    lap = ndi.correlate(g.astype(np.int32), np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.int32), mode="mirror")
    assert np.array_equal(R.laplacian_i32(g), lap)                     # cv2.Laplacian ksize 1, BORDER_REFLECT_101
    dx, dy = R.sobel3_i32(g)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int32)
    assert np.array_equal(dx, ndi.correlate(g.astype(np.int32), kx, mode="nearest"))      # Canny's Sobel: BORDER_REPLICATE
    assert np.array_equal(dy, ndi.correlate(g.astype(np.int32), kx.T, mode="nearest"))


def test_gaussian5_equals_separable_binomial_filter():
    g = _gray(2).astype(np.float32)
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16.0
    want = ndi.correlate1d(ndi.correlate1d(g.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    got = R.gaussian5_f32(g)
    assert got.dtype == np.float32 and np.abs(got - want).max() <= 2e-4     # float32 accumulation vs float64


def test_hsv_within_one_level_of_colorsys():
    rs = np.random.RandomState(3)
    px = rs.randint(0, 256, (1, 4000, 3)).astype(np.uint8)
    hsv = R.bgr2hsv_u8(px)[0].astype(int)
    for (b, g, r), (h, s, v) in zip(px[0][:1500].astype(float), hsv[:1500]):
        hf, sf, vf = colorsys.rgb_to_hsv(r / 255.0, g / 255.0, b / 255.0)
        assert abs(v - vf * 255.0) <= 0.5 and abs(s - sf * 255.0) <= 1.0
        if sf * vf > 0.08:                                                    # hue is ill-conditioned near the grey axis
            dh = abs(h - hf * 180.0)
            assert min(dh, 180.0 - dh) <= 1.5, (b, g, r, h, hf * 180.0)


def _cie_lab(bgr):
    """float64 sRGB (D65) -> CIE L*a*b* with the constants OpenCV documents for COLOR_BGR2Lab, 8-bit encoding"""
    c = bgr[..., ::-1].astype(np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ m.T / np.array([0.950456, 1.0, 1.088754])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = np.where(xyz[..., 1] > 0.008856, 116.0 * f[..., 1] - 16.0, 903.3 * xyz[..., 1])
    return np.stack([L * 255.0 / 100.0, 500.0 * (f[..., 0] - f[..., 1]) + 128.0, 200.0 * (f[..., 1] - f[..., 2]) + 128.0], -1)


def test_lab_forward_within_one_level_of_the_cie_definition_and_inverse_returns():
    rs = np.random.RandomState(4)
    px = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    lab = R.bgr2lab_u8(px).astype(np.float64)
    d = np.abs(lab - np.clip(_cie_lab(px), 0, 255))
    # 8-bit rounding + the 3-bit gamma / 15-bit cube-root tables (the a channel carries a factor 500): a level or two at most
    assert d[..., 0].max() <= 1.25 and d[..., 1:].max() <= 2.0 and d.mean() < 0.4, (d.reshape(-1, 3).max(0), d.mean())
    # inverse: from the float Lab of a colour (rounded to 8 bits) back to within a few levels of that colour
    back = R.lab2bgr_u8(np.clip(np.rint(_cie_lab(px)), 0, 255).astype(np.uint8)).astype(int)
    err = np.abs(back - px.astype(int))
    assert err.mean() < 1.2 and np.percentile(err, 99) <= 8


def test_gray_is_the_601_luma():
    px = np.random.RandomState(5).randint(0, 256, (1, 5000, 3)).astype(np.uint8)
    g = R.bgr2gray_u8(px)[0].astype(np.float64)
    want = px[0].astype(np.float64) @ np.array([0.114, 0.587, 0.299])
    assert np.abs(g - want).max() <= 0.51


def test_resize_linear_within_one_level_of_float_bilinear():
    """cv2.resize(INTER_LINEAR) is plain bilinear sampling at half-pixel centres (no antialiasing) in 11-bit fixed point:
    torch's F.interpolate(mode="bilinear", align_corners=False) is the same sampling in float."""
    import torch
    import torch.nn.functional as F

    rs = np.random.RandomState(6)
    for (h, w, dh, dw) in ((120, 160, 64, 64), (97, 131, 256, 256), (300, 300, 112, 75)):
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = R.resize_linear_u8(img, dw, dh).astype(np.float64)
        t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
        want = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - want).max() <= 1.0, (h, w, dh, dw, np.abs(got - want).max())
