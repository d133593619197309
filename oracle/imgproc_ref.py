"""ORACLE (test infrastructure only): numpy restatement of the OpenCV image operations the
reference's hot path calls.  cv2 is NOT installed in the build container, so these follow
OpenCV's published 8-bit fixed-point algorithms (imgproc color/resize/clahe/canny sources),
restated from their definitions; equality with a real cv2 build is UNVERIFIED -> every use of
this module is "parity unpinned" unless a test says otherwise (the JPEG round trip in
forensics_ref.py IS pinned: it is checked against Pillow's libjpeg).

Reference call sites:
  resize_linear_u8   cv2.resize(..., INTER_LINEAR)      frame_analysis.py:71,112; face_detection.py:77
  bgr2gray_u8        cv2.cvtColor(BGR2GRAY)             frame_analysis.py:136,188,241,286,356
  bgr2hsv_u8         cv2.cvtColor(BGR2HSV)              frame_analysis.py:318
  bgr2lab_u8/lab2bgr_u8, clahe_u8                       deepfake_detection.py:363-368
  gaussian5_f32      cv2.GaussianBlur(gray,(5,5),0)     frame_analysis.py:191
  laplacian_i32      cv2.Laplacian(gray, CV_64F)        frame_analysis.py:293
  canny_u8           cv2.Canny(gray, 50, 150)           frame_analysis.py:289
  crop_resize_normalize  F.interpolate + /255 + normalize   deepfake_detection.py:382-389

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- resize
_COEF_BITS = 11
_COEF_ONE = 1 << _COEF_BITS


def _rint_short(x):
    """saturate_cast<short>(float): round half to even, clamp to int16."""
    return np.clip(np.rint(x), -32768, 32767).astype(np.int32)


def _linear_taps(src_n: int, dst_n: int, clamp_coef: bool):
    scale = float(src_n) / float(dst_n)                       # double
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)          # (float)((dx+0.5)*scale_x - 0.5)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_coef:                                            # x direction: coefficient zeroed at the borders
        lo = s < 0
        f = np.where(lo, np.float32(0), f)
        s = np.where(lo, 0, s)
        hi = s >= src_n - 1
        f = np.where(hi, np.float32(0), f)
        s = np.where(hi, src_n - 1, s)
        s1 = np.minimum(s + 1, src_n - 1)
        s0 = s
    else:                                                     # y direction: row indices clipped instead
        s0 = np.clip(s, 0, src_n - 1)
        s1 = np.clip(s + 1, 0, src_n - 1)
    a0 = _rint_short((np.float32(1.0) - f) * np.float32(_COEF_ONE))
    a1 = _rint_short(f * np.float32(_COEF_ONE))
    return s0, s1, a0, a1


def resize_linear_u8(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    """cv2.resize(src, (dst_w, dst_h), interpolation=INTER_LINEAR) for 8-bit images
    (fixed-point 11-bit coefficients, HResizeLinear + VResizeLinear)."""
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    h, w = src.shape[:2]
    if (w, h) == (dst_w, dst_h):
        out = src.copy()
        return out[:, :, 0] if squeeze else out
    x0, x1, ax0, ax1 = _linear_taps(w, dst_w, True)
    y0, y1, by0, by1 = _linear_taps(h, dst_h, False)
    s = src.astype(np.int32)
    rows0 = s[y0]                                             # (dst_h, w, c)
    rows1 = s[y1]
    h0 = rows0[:, x0] * ax0[None, :, None] + rows0[:, x1] * ax1[None, :, None]
    h1 = rows1[:, x0] * ax0[None, :, None] + rows1[:, x1] * ax1[None, :, None]
    b0 = by0[:, None, None]
    b1 = by1[:, None, None]
    out = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


# --------------------------------------------------------------------------- gray / hsv
def bgr2gray_u8(bgr: np.ndarray) -> np.ndarray:
    b = bgr[..., 0].astype(np.int32)
    g = bgr[..., 1].astype(np.int32)
    r = bgr[..., 2].astype(np.int32)
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


_HSV_SHIFT = 12


def _hsv_tables():
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    sdiv[1:] = np.rint((255 << _HSV_SHIFT) / i)
    hdiv[1:] = np.rint((180 << _HSV_SHIFT) / (6.0 * i))
    return sdiv, hdiv


def bgr2hsv_u8(bgr: np.ndarray) -> np.ndarray:
    """8-bit BGR->HSV with H in [0,180) (OpenCV RGB2HSV_b integer path)."""
    sdiv, hdiv = _hsv_tables()
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * sdiv[v] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << (_HSV_SHIFT - 1))) >> _HSV_SHIFT
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


# --------------------------------------------------------------------------- Lab
# OpenCV modules/imgproc/src/color_lab.cpp (4.x): RGB2Lab_b and Lab2RGBinteger with their tables as initLabTabs()
# builds them.  OpenCV computes the tables in softfloat (IEEE binary32, round to nearest even: numpy float32 gives the
# same +,-,*,/ bit for bit) and the two gamma curves in softdouble; where it calls its own pow / cbrt routines numpy's
# are used (last-ulp differences possible, each rounded to an integer table entry afterwards).
_LAB_SHIFT = 12                     # lab_shift
_GAMMA_SHIFT = 3                    # gamma_shift
_LAB_SHIFT2 = _LAB_SHIFT + _GAMMA_SHIFT
_CBRT_TAB = 256 * 3 // 2 * (1 << _GAMMA_SHIFT)          # LAB_CBRT_TAB_SIZE_B
_D65 = (0.950456, 1.0, 1.088754)
_RGB2XYZ = (0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227)
_XYZ2RGB = (3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311)
_BASE_SHIFT = 14                    # Lab2RGBinteger::base_shift
_BASE = 1 << _BASE_SHIFT            # BASE == LAB_BASE
_INV_GAMMA_SHIFT = 12               # inv_gamma_shift
_INV_GAMMA_TAB = 1 << _INV_GAMMA_SHIFT                  # INV_GAMMA_TAB_SIZE
_INV_SHIFT = _LAB_SHIFT + (_BASE_SHIFT - _INV_GAMMA_SHIFT)   # Lab2RGBinteger::shift == 14
_AB_MIN = -8145                     # minABvalue
_AB_TAB = _BASE * 9 // 4            # 36864 entries of abToXZ_b

_f32 = np.float32


def _cdiv(a, b):
    """C integer division (truncation toward zero) for numpy int64 arrays, b > 0."""
    a = np.asarray(a, np.int64)
    return np.where(a >= 0, a // b, -((-a) // b))


def _fma32(a, b, c):
    """softfloat mulAdd on binary32 values: the product of two float32 is exact in float64."""
    return (np.asarray(a, np.float64) * np.float64(b) + np.float64(c)).astype(np.float32)


def lab_tables() -> dict:
    """Every LUT of the 8-bit BGR<->Lab paths (initLabTabs, RGB2Lab_b / Lab2RGBinteger constructors)."""
    t = {}
    # sRGBGammaTab_b[i] = cvRound(255*(1<<gamma_shift) * applyGamma(i/255)); applyGamma evaluates in double
    x = (np.arange(256, dtype=_f32) / _f32(255)).astype(np.float64)
    lin = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4).astype(_f32)
    t["gamma"] = np.rint(_f32(255 * (1 << _GAMMA_SHIFT)) * lin).astype(np.int32)
    # LabCbrtTab_b[i] = cvRound((1<<lab_shift2) * (x < 216/24389 ? mulAdd(x, 841/108, 16/116) : cbrt(x))), x = i/(255*8)
    y = (_f32(1) / _f32(255 * (1 << _GAMMA_SHIFT))) * np.arange(_CBRT_TAB, dtype=_f32)
    lthresh, lscale, lbias = _f32(216) / _f32(24389), _f32(841) / _f32(108), _f32(16) / _f32(116)
    f = np.where(y < lthresh, _fma32(y, lscale, lbias), np.cbrt(y.astype(np.float64)).astype(_f32)).astype(_f32)
    t["cbrt"] = np.rint(_f32(1 << _LAB_SHIFT2) * f).astype(np.int32)
    # RGB2Lab_b: coeffs[i*3+j] = cvRound((1<<lab_shift) * sRGB2XYZ_D65[i*3+j] / D65[i]) in double
    t["fwd_coef"] = np.array([int(np.rint((1 << _LAB_SHIFT) * _RGB2XYZ[i * 3 + j] / _D65[i]))
                              for i in range(3) for j in range(3)], np.int32)      # rows X,Y,Z ; cols R,G,B
    # ---- inverse path ----
    # LabToYF_b: L (0..255 for 0..100) -> y and f(y), both scaled by BASE
    i = np.arange(256)
    fi = i.astype(_f32)
    y_lo = np.rint((fi * _f32(_BASE * 20 * 9)) / _f32(17 * 29 * 29 * 29))        # y = L*100/255 / (29/3)^3
    # softfloat(i*BASE*20*9): the int product is below 2^24 for i <= 20, so int->float is exact and == fi * const
    fy_lo = np.rint(_f32(_BASE) * (_f32(16) / _f32(116) + (fi * _f32(5)) / _f32(3 * 17 * 29)))
    fy = (i * 100 * _BASE).astype(_f32) / _f32(255 * 116) + _f32(16 * _BASE) / _f32(116)
    y_hi = np.rint(fy * fy * fy / _f32(_BASE * _BASE))
    dark = i <= 20                                            # 8 * 255 / 100 == 20.4
    t["L_y"] = np.where(dark, y_lo, y_hi).astype(np.int32)
    t["L_fy"] = np.where(dark, fy_lo, np.rint(fy)).astype(np.int32)
    # process(): adiv = ((5*aa*53687 + (1 << 7)) >> 13) - 128*BASE/500;  bdiv = ((bb*41943 + (1 << 4)) >> 9) - 128*BASE/200 + 1
    ab = np.arange(256, dtype=np.int64)
    t["a_div"] = (((5 * ab * 53687 + (1 << 7)) >> 13) - 128 * _BASE // 500).astype(np.int32)
    t["b_div"] = (((ab * 41943 + (1 << 4)) >> 9) - 128 * _BASE // 200 + 1).astype(np.int32)
    # abToXZ_b (initLUTforABXZ), C int arithmetic: i <= 3390 (6/29*BASE): i*108/841 - BASE*16/116*108/841, else i*i/BASE*i/BASE
    v = np.arange(_AB_TAB, dtype=np.int64) + _AB_MIN
    lin_part = _cdiv(v * 108, 841) - (_BASE * 16 // 116 * 108 // 841)
    cube_part = _cdiv(_cdiv(v * v, _BASE) * v, _BASE)
    t["ab_xz"] = np.where(v <= 3390, lin_part, cube_part).astype(np.int32)
    # Lab2RGBinteger(): coeffs = cvRound((1<<lab_shift) * XYZ2sRGB_D65[i + j*3]... * whitePt[col]) in double
    t["inv_coef"] = np.array([int(np.rint((1 << _LAB_SHIFT) * _XYZ2RGB[i * 3 + j] * _D65[j]))
                              for i in range(3) for j in range(3)], np.int32)      # rows R,G,B ; cols X,Y,Z
    # sRGBInvGammaTab_b[i] = cvRound(255 * applyInvGamma(i / INV_GAMMA_TAB_SIZE)); applyInvGamma evaluates in double
    xv = ((_f32(1) / _f32(_INV_GAMMA_TAB)) * np.arange(_INV_GAMMA_TAB, dtype=_f32)).astype(np.float64)
    g = np.where(xv <= 0.0031308, xv * 12.92, (xv ** (1.0 / 2.4)) * 1.055 - 0.055).astype(_f32)
    t["inv_gamma"] = np.rint(_f32(255) * g).astype(np.int32)
    return t


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def bgr2lab_u8(bgr: np.ndarray, t=None) -> np.ndarray:
    """8-bit BGR->Lab, OpenCV RGB2Lab_b integer path (gamma LUT, 12-bit matrix, cbrt LUT)."""
    t = t or lab_tables()
    c = t["fwd_coef"].astype(np.int64)
    B = t["gamma"][bgr[..., 0]].astype(np.int64)
    G = t["gamma"][bgr[..., 1]].astype(np.int64)
    R = t["gamma"][bgr[..., 2]].astype(np.int64)
    fX = t["cbrt"][_descale(R * c[0] + G * c[1] + B * c[2], _LAB_SHIFT)].astype(np.int64)
    fY = t["cbrt"][_descale(R * c[3] + G * c[4] + B * c[5], _LAB_SHIFT)].astype(np.int64)
    fZ = t["cbrt"][_descale(R * c[6] + G * c[7] + B * c[8], _LAB_SHIFT)].astype(np.int64)
    Lscale = (116 * 255 + 50) // 100
    Lshift = -((16 * 255 * (1 << _LAB_SHIFT2) + 50) // 100)
    L = _descale(Lscale * fY + Lshift, _LAB_SHIFT2)
    a = _descale(500 * (fX - fY) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    b = _descale(200 * (fY - fZ) + 128 * (1 << _LAB_SHIFT2), _LAB_SHIFT2)
    return np.clip(np.stack([L, a, b], axis=-1), 0, 255).astype(np.uint8)


def lab2bgr_u8(lab: np.ndarray, t=None) -> np.ndarray:
    """8-bit Lab->BGR: OpenCV's Lab2RGBinteger::process (color_lab.cpp), sRGB branch.  y and f(y) from LabToYF_b,
    f(x) = f(y) + adiv, f(z) = f(y) - bdiv, x / z from abToXZ_b, 12-bit matrix, descale by `shift` (14) to a
    12-bit linear value, sRGBInvGammaTab_b."""
    t = t or lab_tables()
    c = t["inv_coef"].astype(np.int64)
    fy = t["L_fy"][lab[..., 0]].astype(np.int64)
    y = t["L_y"][lab[..., 0]].astype(np.int64)
    fx = fy + t["a_div"][lab[..., 1]]
    fz = fy - t["b_div"][lab[..., 2]]
    assert fx.min() >= _AB_MIN and fz.min() >= _AB_MIN and max(fx.max(), fz.max()) < _AB_TAB + _AB_MIN   # OpenCV indexes unchecked
    x = t["ab_xz"][fx - _AB_MIN].astype(np.int64)
    z = t["ab_xz"][fz - _AB_MIN].astype(np.int64)
    out = []
    for row in (2, 1, 0):                                     # B, G, R
        lin = _descale(c[row * 3] * x + c[row * 3 + 1] * y + c[row * 3 + 2] * z, _INV_SHIFT)
        out.append(t["inv_gamma"][np.clip(lin, 0, _INV_GAMMA_TAB - 1)])
    return np.stack(out, axis=-1).astype(np.uint8)


# --------------------------------------------------------------------------- CLAHE
def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def clahe_u8(src: np.ndarray, clip_limit: float = 2.0, tiles=(8, 8)) -> np.ndarray:
    """cv2.createCLAHE(clipLimit, tileGridSize).apply(src) for 8-bit single channel."""
    tx, ty = tiles
    h, w = src.shape
    if h % ty or w % tx:
        eh, ew = h + (ty - h % ty), w + (tx - w % tx)        # OpenCV pads by a full (tiles - rem)
        ys = _reflect101(np.arange(eh), h)
        xs = _reflect101(np.arange(ew), w)
        ext = src[ys][:, xs]
    else:
        ext = src
    th, tw = ext.shape[0] // ty, ext.shape[1] // tx
    area = th * tw
    lut_scale = np.float32(255.0) / np.float32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    luts = np.zeros((ty, tx, 256), np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch, residual = clipped // 256, clipped % 256
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    hist[idx] += 1
            cdf = np.cumsum(hist).astype(np.float32)
            luts[j, i] = np.clip(np.rint(cdf * lut_scale), 0, 255).astype(np.uint8)
    inv_th, inv_tw = np.float32(1.0) / np.float32(th), np.float32(1.0) / np.float32(tw)
    yf = np.arange(h, dtype=np.float32) * inv_th - np.float32(0.5)
    y1 = np.floor(yf).astype(np.int32)
    ya = (yf - y1.astype(np.float32)).astype(np.float32)
    y2 = np.minimum(y1 + 1, ty - 1)
    y1 = np.maximum(y1, 0)
    xf = np.arange(w, dtype=np.float32) * inv_tw - np.float32(0.5)
    x1 = np.floor(xf).astype(np.int32)
    xa = (xf - x1.astype(np.float32)).astype(np.float32)
    x2 = np.minimum(x1 + 1, tx - 1)
    x1 = np.maximum(x1, 0)
    v = src.astype(np.int64)
    Y1, Y2, X1, X2 = y1[:, None], y2[:, None], x1[None, :], x2[None, :]
    l11 = luts[Y1, X1, v].astype(np.float32)
    l12 = luts[Y1, X2, v].astype(np.float32)
    l21 = luts[Y2, X1, v].astype(np.float32)
    l22 = luts[Y2, X2, v].astype(np.float32)
    xa_ = xa[None, :]
    xa1 = (np.float32(1.0) - xa)[None, :]
    ya_ = ya[:, None]
    ya1 = (np.float32(1.0) - ya)[:, None]
    res = (l11 * xa1 + l12 * xa_) * ya1 + (l21 * xa1 + l22 * xa_) * ya_
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def preprocess_face_quality(bgr: np.ndarray) -> np.ndarray:
    """reference deepfake_detection.py:357-370: BGR->Lab, CLAHE(2.0, 8x8) on L, Lab->BGR."""
    t = lab_tables()
    lab = bgr2lab_u8(bgr, t)
    lab[..., 0] = clahe_u8(lab[..., 0], 2.0, (8, 8))
    return lab2bgr_u8(lab, t)


# --------------------------------------------------------------------------- filters
def gaussian5_f32(img: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(img_f32, (5,5), 0): separable [1,4,6,4,1]/16, BORDER_REFLECT_101,
    float32 accumulation in OpenCV's symmetric-filter order (centre, then +-1, then +-2)."""
    k0, k1, k2 = np.float32(0.375), np.float32(0.25), np.float32(0.0625)
    a = img.astype(np.float32)
    h, w = a.shape
    xi = [_reflect101(np.arange(w) + d, w) for d in (-2, -1, 0, 1, 2)]
    r = k0 * a[:, xi[2]]
    r = r + k1 * (a[:, xi[1]] + a[:, xi[3]])
    r = r + k2 * (a[:, xi[0]] + a[:, xi[4]])
    yi = [_reflect101(np.arange(h) + d, h) for d in (-2, -1, 0, 1, 2)]
    o = k0 * r[yi[2]]
    o = o + k1 * (r[yi[1]] + r[yi[3]])
    o = o + k2 * (r[yi[0]] + r[yi[4]])
    return o.astype(np.float32)


def laplacian_i32(gray: np.ndarray) -> np.ndarray:
    """cv2.Laplacian(gray_u8, CV_64F), ksize 1: [0 1 0; 1 -4 1; 0 1 0], BORDER_REFLECT_101 (exact integers)."""
    g = gray.astype(np.int32)
    h, w = g.shape
    up = g[_reflect101(np.arange(h) - 1, h)]
    dn = g[_reflect101(np.arange(h) + 1, h)]
    lf = g[:, _reflect101(np.arange(w) - 1, w)]
    rt = g[:, _reflect101(np.arange(w) + 1, w)]
    return up + dn + lf + rt - 4 * g


def sobel3_i32(gray: np.ndarray):
    """3x3 Sobel dx, dy with BORDER_REPLICATE (what cv2.Canny uses internally)."""
    g = np.pad(gray.astype(np.int32), 1, mode="edge")
    dx = (g[:-2, 2:] + 2 * g[1:-1, 2:] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[1:-1, :-2] + g[2:, :-2])
    dy = (g[2:, :-2] + 2 * g[2:, 1:-1] + g[2:, 2:]) - (g[:-2, :-2] + 2 * g[:-2, 1:-1] + g[:-2, 2:])
    return dx, dy


def canny_u8(gray: np.ndarray, low: int = 50, high: int = 150) -> np.ndarray:
    """cv2.Canny(gray, low, high): aperture 3, L1 magnitude, fixed-point non-maximum suppression
    (tan 22.5 deg = 13573/2^15), 8-connected hysteresis.  Returns 0/255."""
    dx, dy = sobel3_i32(gray)
    mag = np.abs(dx) + np.abs(dy)
    h, w = mag.shape
    mp = np.pad(mag, 1)                                        # zero border like OpenCV's mag buffer
    m = mp[1:-1, 1:-1]
    TG22 = 13573
    x = np.abs(dx).astype(np.int64)
    y = np.abs(dy).astype(np.int64) << 15
    tg22x = x * TG22
    tg67x = tg22x + (x << 16)
    left, right = mp[1:-1, :-2], mp[1:-1, 2:]
    up, down = mp[:-2, 1:-1], mp[2:, 1:-1]
    s_neg = (dx ^ dy) < 0                                      # gradient along the anti-diagonal
    # s = -1: compare prev[j+1], next[j-1] ; s = +1: prev[j-1], next[j+1]
    diag_prev = np.where(s_neg, mp[:-2, 2:], mp[:-2, :-2])
    diag_next = np.where(s_neg, mp[2:, :-2], mp[2:, 2:])
    horiz = (y < tg22x) & (m > left) & (m >= right)
    vert = (y > tg67x) & (m > up) & (m >= down)
    diag = (y >= tg22x) & (y <= tg67x) & (m > diag_prev) & (m > diag_next)
    cand = (m > low) & (horiz | vert | diag)
    strong = cand & (m > high)
    # hysteresis: grow `strong` through 8-connected candidates
    edge = strong.copy()
    stack = list(zip(*np.nonzero(strong)))
    while stack:
        i, j = stack.pop()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                a, b = i + di, j + dj
                if 0 <= a < h and 0 <= b < w and cand[a, b] and not edge[a, b]:
                    edge[a, b] = True
                    stack.append((a, b))
    return (edge * 255).astype(np.uint8)


# --------------------------------------------------------------------------- crop -> network input
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def crop_resize_normalize(face_bgr: np.ndarray) -> np.ndarray:
    """reference deepfake_detection.py:376,382-389 with the MTCNN re-crop bypassed:
    BGR u8 crop -> RGB float 0..255 -> F.interpolate(224, bilinear, align_corners=False) -> /255
    -> (x-mean)/std.  Returns (3,224,224) float32."""
    import torch
    import torch.nn.functional as F

    rgb = torch.from_numpy(np.ascontiguousarray(face_bgr[..., ::-1])).permute(2, 0, 1).float().unsqueeze(0)
    x = F.interpolate(rgb, size=(224, 224), mode="bilinear", align_corners=False)
    x = x.to(torch.float32) / 255.0
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return ((x - mean) / std)[0].numpy()


def compute_frequency_features(image: np.ndarray, size: int = 224) -> np.ndarray:
    """reference model.py:105-149: gray -> cv2.resize(size,size) -> float32; channel 0 = min-max of
    log1p|fftshift(fft2)|, channel 1 = min-max of log1p|cv2.dct(gray/255)| (orthonormal DCT-II, here
    via scipy.fft.dctn(type=2, norm='ortho'))."""
    from scipy.fft import dctn

    gray = bgr2gray_u8(image) if image.ndim == 3 else image
    gray = resize_linear_u8(gray, size, size).astype(np.float32)

    def norm(a):
        lo, hi = a.min(), a.max()
        return (a - lo) / (hi - lo) if hi - lo > 1e-6 else np.zeros_like(a)

    mag = norm(np.log1p(np.abs(np.fft.fftshift(np.fft.fft2(gray)))))
    d = norm(np.log1p(np.abs(dctn(gray / np.float32(255.0), type=2, norm="ortho"))))
    return np.stack([mag, d], axis=0).astype(np.float32)
