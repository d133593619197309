"""Integer lookup tables of the 8-bit colour conversions, generated on the host and shipped
to the GPU inside the weights blob (as exactly-representable float32 values; the C side
converts them to int32 device tables at `dfd_create`).

The per-pixel arithmetic of BGR->Lab, Lab->BGR and BGR->HSV on the GPU is integer/LUT only
(OpenCV's 8-bit fixed-point formulation, which is what reference
deepfake_detection.py:363-368 and frame_analysis.py:318 execute through cv2), so the kernels
are bit-exact against the oracle by construction.  `tests/test_imgproc.py` checks these tables
against the oracle's independent construction.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

GAMMA_SHIFT = 3
LAB_SHIFT = 12
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
CBRT_TAB_SIZE = 256 * 3 // 2 * (1 << GAMMA_SHIFT)        # 3072
INV_BITS = 14
INV_ONE = 1 << INV_BITS
AB_MIN = -8145
AB_TAB_SIZE = INV_ONE * 9 // 4                           # 36864
HSV_SHIFT = 12

_WHITE = np.array([0.950456, 1.0, 1.088754])
_M_FWD = np.array([[0.412453, 0.357580, 0.180423],
                   [0.212671, 0.715160, 0.072169],
                   [0.019334, 0.119193, 0.950227]])
_M_INV = np.array([[3.240479, -1.53715, -0.498535],
                   [-0.969256, 1.875991, 0.041556],
                   [0.055648, -0.204043, 1.057311]])


def _srgb_to_linear(v):
    v = np.asarray(v, np.float64)
    return np.where(v <= 0.04045, v / 12.92, np.power((v + 0.055) / 1.055, 2.4))


def _linear_to_srgb(v):
    v = np.asarray(v, np.float64)
    return np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1 / 2.4) - 0.055)


def _lab_f(t):
    t = np.asarray(t, np.float64)
    return np.where(t < 0.008856, 7.787 * t + 16.0 / 116.0, np.cbrt(t))


def build() -> Dict[str, np.ndarray]:
    """name -> int64 array; every value is < 2^24 in magnitude (exact in float32)."""
    t: Dict[str, np.ndarray] = {}
    t["lut.gamma"] = np.rint(255.0 * 8.0 * _srgb_to_linear(np.arange(256) / 255.0)).astype(np.int64)
    t["lut.cbrt"] = np.rint(32768.0 * _lab_f(np.arange(CBRT_TAB_SIZE) / (255.0 * 8.0))).astype(np.int64)
    t["lut.fwd_coef"] = np.rint(4096.0 * _M_FWD / _WHITE[:, None]).astype(np.int64).ravel()
    L = np.arange(256) * (100.0 / 255.0)
    dark = L <= 8.0
    y = np.where(dark, L / 903.3, ((L + 16.0) / 116.0) ** 3)
    fy = np.where(dark, 7.787 * y + 16.0 / 116.0, (L + 16.0) / 116.0)
    t["lut.L_fy"] = np.rint(fy * INV_ONE).astype(np.int64)
    t["lut.L_y"] = np.rint(y * INV_ONE).astype(np.int64)
    ab = np.arange(256) - 128.0
    t["lut.a_div"] = np.rint(ab * (INV_ONE / 500.0)).astype(np.int64)
    t["lut.b_div"] = np.rint(ab * (INV_ONE / 200.0)).astype(np.int64)
    f = (np.arange(AB_TAB_SIZE) + AB_MIN) / float(INV_ONE)
    t["lut.ab_xz"] = np.rint(np.where(f <= 6.0 / 29.0, (f - 16.0 / 116.0) / 7.787, f ** 3) * INV_ONE).astype(np.int64)
    t["lut.inv_coef"] = np.rint(4096.0 * _M_INV * _WHITE[None, :]).astype(np.int64).ravel()
    t["lut.inv_gamma"] = np.clip(np.rint(255.0 * _linear_to_srgb(np.arange(INV_ONE + 1) / float(INV_ONE))), 0, 255).astype(np.int64)
    i = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256)
    hdiv = np.zeros(256)
    sdiv[1:] = np.rint((255 << HSV_SHIFT) / i)
    hdiv[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * i))
    t["lut.hsv_sdiv"] = sdiv.astype(np.int64)
    t["lut.hsv_hdiv"] = hdiv.astype(np.int64)
    for k, v in t.items():
        assert np.abs(v).max() < (1 << 24), k
    return t


def as_float_tensors() -> Dict[str, np.ndarray]:
    return {k: v.astype(np.float32) for k, v in build().items()}
