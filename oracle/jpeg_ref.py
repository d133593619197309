"""ORACLE (test infrastructure only): the JPEG encode->decode round trip behind the reference's
Error Level Analysis (cv2.imencode('.jpg', frame, [IMWRITE_JPEG_QUALITY, 90]) then
cv2.imdecode, reference frame_analysis.py:233-236), restated in numpy from libjpeg's integer
algorithms: jccolor.c (RGB->YCbCr), jcsample.c (h2v2 downsample), jfdctint.c (islow FDCT),
jcdctmgr.c (quantise), jdcoefct/jidctint.c (dequantise + islow IDCT), jdsample.c (h2v2 "fancy"
triangle upsample), jdcolor.c (YCbCr->RGB).  The Huffman stage is lossless and is skipped.

PINNED: `roundtrip_pil` runs the same round trip through Pillow's libjpeg(-turbo) - the codec
family OpenCV links - and tests/test_jpeg_ref.py requires `roundtrip_np` to reproduce it
byte for byte.  (cv2 itself is absent from the build container; that its imencode defaults are
quality-scaled Annex-K tables, 4:2:0, islow, fancy upsampling is taken from OpenCV's source.)

Only for image sizes that are multiples of 16 (the forensic path always uses 256x256).
"""
from __future__ import annotations

import io

import numpy as np

_LUMA = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                  14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                  49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], np.int64).reshape(8, 8)
_CHROMA = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                    47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, np.int64).reshape(8, 8)

_C = dict(F0298=2446, F0390=3196, F0541=4433, F0765=6270, F0899=7373, F1175=9633, F1501=12299, F1847=15137,
          F1961=16069, F2053=16819, F2562=20995, F3072=25172)
_CONST_BITS, _PASS1_BITS = 13, 2


def quant_table(base: np.ndarray, quality: int) -> np.ndarray:
    """jpeg_quality_scaling + jpeg_add_quant_table(force_baseline=TRUE)."""
    quality = min(max(quality, 1), 100)
    scale = 5000 // quality if quality < 50 else 200 - quality * 2
    return np.clip((base * scale + 50) // 100, 1, 255)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fix(x):
    return int(x * 65536 + 0.5)


def rgb_to_ycc(rgb: np.ndarray):
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    half, off = 1 << 15, 128 << 16
    y = (_fix(0.29900) * r + _fix(0.58700) * g + _fix(0.11400) * b + half) >> 16
    cb = (-_fix(0.16874) * r - _fix(0.33126) * g + _fix(0.50000) * b + off + half - 1) >> 16
    cr = (_fix(0.50000) * r - _fix(0.41869) * g - _fix(0.08131) * b + off + half - 1) >> 16
    return y, cb, cr


def h2v2_downsample(p: np.ndarray) -> np.ndarray:
    s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(s.shape[1]) % 2 == 0, 1, 2)[None, :]
    return (s + bias) >> 2


def _dct_1d(d, first_pass: bool):
    """One pass of jpeg_fdct_islow over the last axis of d (..., 8)."""
    c = _C
    t0, t7 = d[..., 0] + d[..., 7], d[..., 0] - d[..., 7]
    t1, t6 = d[..., 1] + d[..., 6], d[..., 1] - d[..., 6]
    t2, t5 = d[..., 2] + d[..., 5], d[..., 2] - d[..., 5]
    t3, t4 = d[..., 3] + d[..., 4], d[..., 3] - d[..., 4]
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    out = np.empty_like(d)
    if first_pass:
        out[..., 0] = (t10 + t11) << _PASS1_BITS
        out[..., 4] = (t10 - t11) << _PASS1_BITS
        n = _CONST_BITS - _PASS1_BITS
    else:
        out[..., 0] = _descale(t10 + t11, _PASS1_BITS)
        out[..., 4] = _descale(t10 - t11, _PASS1_BITS)
        n = _CONST_BITS + _PASS1_BITS
    z1 = (t12 + t13) * c["F0541"]
    out[..., 2] = _descale(z1 + t13 * c["F0765"], n)
    out[..., 6] = _descale(z1 + t12 * (-c["F1847"]), n)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * c["F1175"]
    t4, t5, t6, t7 = t4 * c["F0298"], t5 * c["F2053"], t6 * c["F3072"], t7 * c["F1501"]
    z1, z2 = z1 * (-c["F0899"]), z2 * (-c["F2562"])
    z3, z4 = z3 * (-c["F1961"]) + z5, z4 * (-c["F0390"]) + z5
    out[..., 7] = _descale(t4 + z1 + z3, n)
    out[..., 5] = _descale(t5 + z2 + z4, n)
    out[..., 3] = _descale(t6 + z2 + z3, n)
    out[..., 1] = _descale(t7 + z1 + z4, n)
    return out


def fdct_islow(blocks: np.ndarray) -> np.ndarray:
    """blocks (...,8,8) of samples-128 -> coefficients scaled by 8."""
    rows = _dct_1d(blocks, True)
    return np.swapaxes(_dct_1d(np.swapaxes(rows, -1, -2), False), -1, -2)


def quantize(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    qv = q << 3
    a = (np.abs(coef) + (qv >> 1)) // qv
    return np.where(coef < 0, -a, a)


def _idct_1d(v, first_pass: bool):
    c = _C
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * c["F0541"]
    t2 = z1 + z3 * (-c["F1847"])
    t3 = z1 + z2 * c["F0765"]
    t0 = (v[..., 0] + v[..., 4]) << _CONST_BITS
    t1 = (v[..., 0] - v[..., 4]) << _CONST_BITS
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * c["F1175"]
    t0, t1, t2, t3 = t0 * c["F0298"], t1 * c["F2053"], t2 * c["F3072"], t3 * c["F1501"]
    z1, z2 = z1 * (-c["F0899"]), z2 * (-c["F2562"])
    z3, z4 = z3 * (-c["F1961"]) + z5, z4 * (-c["F0390"]) + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    n = _CONST_BITS - _PASS1_BITS if first_pass else _CONST_BITS + _PASS1_BITS + 3
    out = np.empty_like(v)
    out[..., 0], out[..., 7] = _descale(t10 + t3, n), _descale(t10 - t3, n)
    out[..., 1], out[..., 6] = _descale(t11 + t2, n), _descale(t11 - t2, n)
    out[..., 2], out[..., 5] = _descale(t12 + t1, n), _descale(t12 - t1, n)
    out[..., 3], out[..., 4] = _descale(t13 + t0, n), _descale(t13 - t0, n)
    return out


def idct_islow(coef: np.ndarray) -> np.ndarray:
    """dequantised coefficients (...,8,8) -> samples 0..255 (pass 1 on columns, pass 2 on rows)."""
    cols = np.swapaxes(_idct_1d(np.swapaxes(coef, -1, -2), True), -1, -2)
    return np.clip(_idct_1d(cols, False) + 128, 0, 255)


def _blocks(p):
    h, w = p.shape
    return p.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def _unblocks(b):
    nh, nw = b.shape[:2]
    return b.transpose(0, 2, 1, 3).reshape(nh * 8, nw * 8)


def code_plane(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    coef = quantize(fdct_islow(_blocks(p.astype(np.int64) - 128)), q)
    return _unblocks(idct_islow(coef * q))


def h2v2_fancy_upsample(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: 9/16,3/16,3/16,1/16 triangle filter with alternating
    rounding (8 for even, 7 for odd output columns); edge rows/columns replicate."""
    h, w = p.shape
    up = np.concatenate([p[:1], p[:-1]], 0)       # neighbour row for the upper output row
    dn = np.concatenate([p[1:], p[-1:]], 0)       # ... for the lower one
    out = np.empty((2 * h, 2 * w), np.int64)
    for v, nb in ((0, up), (1, dn)):
        colsum = 3 * p + nb                       # (h, w)
        last = np.concatenate([colsum[:, :1], colsum[:, :-1]], 1)
        nxt = np.concatenate([colsum[:, 1:], colsum[:, -1:]], 1)
        even = (3 * colsum + last + 8) >> 4
        odd = (3 * colsum + nxt + 7) >> 4
        even[:, 0] = (4 * colsum[:, 0] + 8) >> 4
        odd[:, -1] = (4 * colsum[:, -1] + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    half = 1 << 15
    xb, xr = cb - 128, cr - 128
    r = y + ((_fix(1.40200) * xr + half) >> 16)
    g = y + ((-_fix(0.34414) * xb + half - _fix(0.71414) * xr) >> 16)
    b = y + ((_fix(1.77200) * xb + half) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def roundtrip_np(bgr: np.ndarray, quality: int = 90) -> np.ndarray:
    """BGR u8 (H,W,3), H,W multiples of 16 -> decoded BGR u8 after a quality-`quality` 4:2:0 round trip."""
    h, w = bgr.shape[:2]
    if h % 16 or w % 16:
        raise ValueError("sizes must be multiples of 16")
    y, cb, cr = rgb_to_ycc(bgr[..., ::-1])
    ql, qc = quant_table(_LUMA, quality), quant_table(_CHROMA, quality)
    y2 = code_plane(y, ql)
    cb2 = h2v2_fancy_upsample(code_plane(h2v2_downsample(cb), qc))
    cr2 = h2v2_fancy_upsample(code_plane(h2v2_downsample(cr), qc))
    return ycc_to_rgb(y2, cb2, cr2)[..., ::-1].copy()


def roundtrip_pil(bgr: np.ndarray, quality: int = 90) -> np.ndarray:
    """The same round trip through Pillow's libjpeg (the pin for roundtrip_np)."""
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr[..., ::-1])).save(buf, format="JPEG", quality=quality)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))[..., ::-1].copy()


# ------------------------------------------------------------------ decoding from coefficients
def h2v1_fancy_upsample(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample: 3/4 + 1/4 with rounding 1 (even) / 2 (odd output columns); ends copy."""
    h, w = p.shape
    out = np.empty((h, 2 * w), np.int64)
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def decode_from_coefficients(info: dict) -> np.ndarray:
    """What libjpeg does after entropy decoding, for the dict `_lib.jpeg_coefficients` returns: dequantise, islow
    IDCT, crop each component plane to its real (downsampled) size, fancy upsample, YCbCr -> RGB.  -> BGR u8."""
    W, H, n = info["width"], info["height"], info["components"]
    planes, off = [], 0
    for c, (bw, bh, tq) in enumerate(info["comps"]):
        cnt = bw * bh * 64
        blk = info["coef"][off:off + cnt].astype(np.int64).reshape(bh, bw, 8, 8)
        off += cnt
        q = info["qtables"][tq].astype(np.int64).reshape(8, 8)
        planes.append(_unblocks(idct_islow(blk * q)))
    y = planes[0][:H, :W]
    if n == 1:
        return np.repeat(y[..., None], 3, -1).astype(np.uint8)
    hmax, vmax = info["hmax"], info["vmax"]
    cw, ch = -(-W // hmax), -(-H // vmax)
    cb, cr = planes[1][:ch, :cw], planes[2][:ch, :cw]
    if hmax == 2 and vmax == 2:
        cb, cr = h2v2_fancy_upsample(cb), h2v2_fancy_upsample(cr)
    elif hmax == 2:
        cb, cr = h2v1_fancy_upsample(cb), h2v1_fancy_upsample(cr)
    return ycc_to_rgb(y, cb[:H, :W], cr[:H, :W])[..., ::-1].copy()
