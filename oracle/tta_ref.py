"""ORACLE (test infrastructure only): the reference's test-time augmentation (deepfake_detection.py:408-443) with the
three cv2 calls restated on numpy - cv2.flip(img, 1); cv2.convertScaleAbs(img, alpha, beta=0) (8-bit: saturate(rint(|v *
float(alpha)|))); cv2.getRotationMatrix2D + cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) in OpenCV's fixed-point form
(imgwarp.cpp: inverse matrix in double, AB_BITS 10, INTER_BITS 5, bilinear weights of 32768, rounding 1 << 14).
PARITY UNPINNED: cv2 is absent here."""
from __future__ import annotations

import math

import numpy as np


def convert_scale_abs(img: np.ndarray, alpha: float) -> np.ndarray:
    v = np.abs(img.astype(np.float32) * np.float32(alpha))
    return np.minimum(np.rint(v), 255).astype(np.uint8)


def inverse_rotation(w: int, h: int, angle_deg: float):
    cx, cy = w / 2.0, h / 2.0
    a, b = math.cos(angle_deg * math.pi / 180.0), math.sin(angle_deg * math.pi / 180.0)
    M = [a, b, (1.0 - a) * cx - b * cy, -b, a, b * cx + (1.0 - a) * cy]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0], M[1], M[3], M[4] = A11, M[1] * -D, M[3] * -D, A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def warp_affine_linear(img: np.ndarray, Mi) -> np.ndarray:
    h, w = img.shape[:2]
    xs, ys = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    ad = np.rint(Mi[0] * xs * 1024.0).astype(np.int64)
    bd = np.rint(Mi[3] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((Mi[1] * ys + Mi[2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((Mi[4] * ys + Mi[5]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + ad[None, :]) >> 5
    Y = (Y0[:, None] + bd[None, :]) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    acc = np.zeros((h, w, img.shape[2]), np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            wgt = (fx if dx else 32 - fx) * (fy if dy else 32 - fy) * 32
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
            acc += np.where(ok[..., None], v, 0) * wgt[..., None]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def augment(img: np.ndarray, flip: bool, brightness: float, angle_deg: float) -> np.ndarray:
    a = img[:, ::-1] if flip else img
    a = convert_scale_abs(np.ascontiguousarray(a), brightness)
    return warp_affine_linear(a, inverse_rotation(img.shape[1], img.shape[0], angle_deg))
