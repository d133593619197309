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
BASE_SHIFT = 14                                          # Lab2RGBinteger::base_shift
BASE = 1 << BASE_SHIFT
INV_GAMMA_SHIFT = 12
INV_GAMMA_TAB_SIZE = 1 << INV_GAMMA_SHIFT                # 4096
INV_SHIFT = LAB_SHIFT + BASE_SHIFT - INV_GAMMA_SHIFT     # 14: descale of the 12-bit matrix product
AB_MIN = -8145                                           # minABvalue
AB_TAB_SIZE = BASE * 9 // 4                              # 36864
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


def _trunc_div(a: int, b: int) -> int:
    """C's integer division."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _f(v) -> np.float32:
    return np.float32(v)


def build() -> Dict[str, np.ndarray]:
    """name -> int64 array; every value is < 2^24 in magnitude (exact in float32).

    The tables are the ones OpenCV's 8-bit Lab conversions use (imgproc/src/color_lab.cpp `initLabTabs`,
    `RGB2Lab_b`, `Lab2RGBinteger`): binary32 arithmetic where OpenCV uses softfloat, the gamma curves in double."""
    t: Dict[str, np.ndarray] = {}
    # forward: sRGBGammaTab_b, LabCbrtTab_b, 12-bit RGB->XYZ/white matrix
    g = _srgb_to_linear((np.arange(256, dtype=np.float32) / _f(255)).astype(np.float64)).astype(np.float32)
    t["lut.gamma"] = np.rint(_f(255 * 8) * g).astype(np.int64)
    x = (_f(1) / _f(255 * 8)) * np.arange(CBRT_TAB_SIZE, dtype=np.float32)
    lin = (x.astype(np.float64) * np.float64(_f(841) / _f(108)) + np.float64(_f(16) / _f(116))).astype(np.float32)   # fused multiply-add
    fx = np.where(x < _f(216) / _f(24389), lin, np.cbrt(x.astype(np.float64)).astype(np.float32))
    t["lut.cbrt"] = np.rint(_f(32768) * fx.astype(np.float32)).astype(np.int64)
    t["lut.fwd_coef"] = np.rint(4096.0 * _M_FWD / _WHITE[:, None]).astype(np.int64).ravel()
    # inverse: LabToYF_b (y and f(y) per L), adiv / bdiv per a / b, abToXZ_b, 12-bit XYZ*white->RGB matrix, sRGBInvGammaTab_b
    L_y, L_fy = [], []
    for i in range(256):
        if i <= 20:                                          # L <= 8
            y = _f(i * BASE * 20 * 9) / _f(17 * 29 * 29 * 29)
            fy = _f(BASE) * (_f(16) / _f(116) + _f(i * 5) / _f(3 * 17 * 29))
        else:
            fy = _f(i * 100 * BASE) / _f(255 * 116) + _f(16 * BASE) / _f(116)
            y = fy * fy * fy / _f(BASE * BASE)
        L_y.append(int(np.rint(y)))
        L_fy.append(int(np.rint(fy)))
    t["lut.L_y"] = np.array(L_y, np.int64)
    t["lut.L_fy"] = np.array(L_fy, np.int64)
    t["lut.a_div"] = np.array([((5 * a * 53687 + 128) >> 13) - 128 * BASE // 500 for a in range(256)], np.int64)
    t["lut.b_div"] = np.array([((b * 41943 + 16) >> 9) - 128 * BASE // 200 + 1 for b in range(256)], np.int64)
    k_lin = BASE * 16 // 116 * 108 // 841
    t["lut.ab_xz"] = np.array([_trunc_div(v * 108, 841) - k_lin if v <= 3390 else v * v // BASE * v // BASE
                               for v in range(AB_MIN, AB_MIN + AB_TAB_SIZE)], np.int64)
    t["lut.inv_coef"] = np.rint(4096.0 * _M_INV * _WHITE[None, :]).astype(np.int64).ravel()
    xs = ((_f(1) / _f(INV_GAMMA_TAB_SIZE)) * np.arange(INV_GAMMA_TAB_SIZE, dtype=np.float32)).astype(np.float64)
    t["lut.inv_gamma"] = np.rint(_f(255) * _linear_to_srgb(xs).astype(np.float32)).astype(np.int64)
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
