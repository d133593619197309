"""CPU: product colour tables vs the oracle's independent construction, and sanity of the
image-processing oracle itself (properties that hold for any correct implementation)."""
import numpy as np

from oracle import imgproc_ref as R


def test_product_tables_equal_oracle_tables(pkg):
    mine = pkg.luts.build()
    ref = R.lab_tables()
    pairs = {"lut.gamma": "gamma", "lut.cbrt": "cbrt", "lut.fwd_coef": "fwd_coef", "lut.L_fy": "L_fy",
             "lut.L_y": "L_y", "lut.a_div": "a_div", "lut.b_div": "b_div", "lut.ab_xz": "ab_xz",
             "lut.inv_coef": "inv_coef", "lut.inv_gamma": "inv_gamma"}
    for a, b in pairs.items():
        assert np.array_equal(mine[a], ref[b].astype(np.int64)), a
    sdiv, hdiv = R._hsv_tables()
    assert np.array_equal(mine["lut.hsv_sdiv"], sdiv) and np.array_equal(mine["lut.hsv_hdiv"], hdiv)


def test_resize_identity_constant_and_shapes():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(R.resize_linear_u8(img, 53, 37), img)
    flat = np.full((1080, 1920, 3), 77, np.uint8)
    assert np.all(R.resize_linear_u8(flat, 256, 256) == 77)
    up = R.resize_linear_u8(img, 300, 300)
    assert up.shape == (300, 300, 3) and up.min() >= img.min() and up.max() <= img.max()
    # 2x upsample of a horizontal ramp stays monotone
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8)[None, :, None], (8, 1, 3))
    r2 = R.resize_linear_u8(ramp, 100, 8).astype(int)
    assert np.all(np.diff(r2[0, :, 0]) >= 0)


def test_gray_hsv_known_values():
    px = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 100]]], np.uint8)
    g = R.bgr2gray_u8(px)[0]
    assert list(g[:5]) == [0, 255, 29, 150, 76]              # ITU-R 601 weights, fixed point
    hsv = R.bgr2hsv_u8(px)[0]
    assert list(hsv[0]) == [0, 0, 0] and list(hsv[1]) == [0, 0, 255]
    assert list(hsv[2]) == [120, 255, 255] and list(hsv[3]) == [60, 255, 255] and list(hsv[4]) == [0, 255, 255]
    assert hsv[:, 0].max() < 180


def test_lab_roundtrip_and_anchors():
    t = R.lab_tables()
    px = np.array([[[0, 0, 0], [255, 255, 255], [128, 128, 128]]], np.uint8)
    lab = R.bgr2lab_u8(px, t)[0]
    assert list(lab[0]) == [0, 128, 128] and list(lab[1]) == [255, 128, 128] and abs(int(lab[2][0]) - 137) <= 1
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    back = R.lab2bgr_u8(R.bgr2lab_u8(img, t), t)
    d = np.abs(back.astype(int) - img.astype(int))
    # 8-bit Lab quantisation only: sub-level on average, a few levels in the dark saturated corner
    assert d.mean() < 1.0 and np.percentile(d, 99) <= 8 and d.max() <= 32


def test_lab2rgb_integer_table_anchors():
    """Facts OpenCV's color_lab.cpp states about its own Lab2RGBinteger tables (source comments and constants): the a/b
    offsets reach down to exactly minABvalue = -8145 over the 8-bit cube, abToXZ_b spans [-1335, 88231] over
    LAB_BASE*9/4 entries, f(y) starts at BASE*16/116, the inverse gamma table has 2^12 entries ending at 255."""
    t = R.lab_tables()
    fx = t["L_fy"][:, None].astype(int) + t["a_div"][None, :]
    fz = t["L_fy"][:, None].astype(int) - t["b_div"][None, :]
    assert min(fx.min(), fz.min()) == -8145 and max(fx.max(), fz.max()) < 36864 - 8145
    assert len(t["ab_xz"]) == 36864 and t["ab_xz"].min() == -1335 and t["ab_xz"].max() == 88231
    assert t["L_fy"][0] == 2260 and t["L_fy"][255] == 16384 and t["L_y"][0] == 0 and t["L_y"][255] == 16384
    assert np.all(np.diff(t["L_y"]) > 0) and np.all(np.diff(t["ab_xz"]) >= 0)
    assert len(t["inv_gamma"]) == 4096 and t["inv_gamma"][0] == 0 and t["inv_gamma"][-1] == 255
    assert t["a_div"][128] == 0 and t["b_div"][128] == 1          # the "+1" of bdiv is OpenCV's, "not a typo" there
    # greys survive the 8-bit round trip within one level
    g = np.arange(256, dtype=np.uint8)
    grey = np.stack([g, g, g], -1)[None]
    assert np.abs(R.lab2bgr_u8(R.bgr2lab_u8(grey, t), t).astype(int) - grey).max() <= 1


def test_clahe_properties():
    rs = np.random.RandomState(2)
    flat = np.full((64, 64), 90, np.uint8)
    out = R.clahe_u8(flat)
    assert out.shape == flat.shape and len(np.unique(out)) == 1
    low = (rs.rand(80, 104) * 40 + 100).astype(np.uint8)      # low-contrast, ragged size (pads)
    eq = R.clahe_u8(low)
    assert eq.shape == low.shape and int(eq.max()) - int(eq.min()) > int(low.max()) - int(low.min())
    # monotone within a single tile-LUT region: equalisation never reorders grey levels of a flat-LUT image
    g = np.tile(np.arange(256, dtype=np.uint8), (256, 1))
    e = R.clahe_u8(g).astype(int)
    assert np.all(np.diff(e[128]) >= -1)


def test_gaussian_laplacian_canny_properties():
    rs = np.random.RandomState(3)
    const = np.full((32, 32), 50, np.uint8)
    assert np.allclose(R.gaussian5_f32(const), 50.0) and np.all(R.laplacian_i32(const) == 0)
    assert R.canny_u8(const).max() == 0
    sq = np.zeros((64, 64), np.uint8)
    sq[16:48, 16:48] = 255
    e = R.canny_u8(sq)
    assert e[16, 30] == 255 or e[15, 30] == 255                # an edge along the square's border
    assert e[32, 32] == 0 and e[2, 2] == 0
    assert set(np.unique(e)) <= {0, 255}
    imp = np.zeros((9, 9), np.float32)
    imp[4, 4] = 256.0
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    assert np.allclose(R.gaussian5_f32(imp)[2:7, 2:7], 256 * np.outer(k, k))


def test_crop_resize_normalize_shape_and_range():
    rs = np.random.RandomState(4)
    face = rs.randint(0, 256, (90, 70, 3)).astype(np.uint8)
    x = R.crop_resize_normalize(face)
    assert x.shape == (3, 224, 224) and x.dtype == np.float32
    assert x.min() >= (0 - 0.485) / 0.229 - 1e-5 and x.max() <= (1 - 0.406) / 0.225 + 1e-5
    same = R.crop_resize_normalize(np.full((224, 224, 3), 128, np.uint8))
    assert np.allclose(same[0], (128 / 255 - 0.485) / 0.229, atol=1e-6)
