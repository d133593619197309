"""Synthetic frames shared by the tests: the reference's inline generators
(tests/test_functional.py:24-39, tests/test_algorithm.py:21-43) restated without cv2."""
import numpy as np


def face_frame(width=640, height=480, seed=0):
    rs = np.random.RandomState(seed)
    f = rs.randint(50, 200, (height, width, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    cx, cy = width // 2, height // 2
    f[((xx - cx) / 80.0) ** 2 + ((yy - cy) / 110.0) ** 2 <= 1.0] = (180, 160, 140)      # skin-toned oval
    for ex in (cx - 30, cx + 30):
        f[(xx - ex) ** 2 + (yy - (cy - 20)) ** 2 <= 64] = (60, 40, 30)                    # "eyes"
    return f


def blank_frame(width=640, height=480):
    return np.full((height, width, 3), 128, np.uint8)


def smooth_image(size=(256, 256)):
    return np.full((*size, 3), 128, np.uint8)             # a blurred constant image is that constant


def noisy_image(size=(256, 256), seed=1):
    return np.random.RandomState(seed).randint(60, 200, (*size, 3)).astype(np.uint8)


def gradient_image(size=(256, 256)):
    h, w = size
    img = np.zeros((h, w, 3), np.uint8)
    img[:, :, :] = (255 * np.arange(h) // h).astype(np.uint8)[:, None, None]
    yy, xx = np.mgrid[0:h, 0:w]
    box = ((np.abs(xx - 50) <= 1) | (np.abs(xx - 200) <= 1)) & (yy >= 49) & (yy <= 201) | \
          ((np.abs(yy - 50) <= 1) | (np.abs(yy - 200) <= 1)) & (xx >= 49) & (xx <= 201)
    img[box] = (255, 0, 0)
    ring = np.abs(np.sqrt((xx - 128.0) ** 2 + (yy - 128.0) ** 2) - 60.0) <= 1.5
    img[ring] = (0, 255, 0)
    return img


def determinism_frame():
    """the reference's own determinism input, tests/test_reliability.py:137"""
    return np.random.RandomState(42).randint(0, 255, (256, 256, 3)).astype(np.uint8)


def natural_like(h=720, w=1280, seed=3):
    """smooth low-frequency content + mild noise: statistics away from every threshold corner"""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 80 * np.sin(xx / 97.0 + c) * np.cos(yy / 61.0 - c) for c in (0.0, 1.0, 2.0)], -1)
    return np.clip(base + rs.randn(h, w, 3) * 6.0, 0, 255).astype(np.uint8)


def varied_frame(i, h=480, w=640):
    """frames whose contrast, brightness, texture scale and noise level change from frame to frame, so that the
    classifier's probabilities spread over a wide range (the bf16 vote gate needs a threshold inside the spread)"""
    rs = np.random.RandomState(900 + i)
    amp, off, sig = rs.uniform(15, 110), rs.uniform(70, 185), rs.uniform(1, 28)
    fx, fy = rs.uniform(40, 160), rs.uniform(30, 120)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([off + amp * np.sin(xx / fx + c) * np.cos(yy / fy - c) for c in (0.0, 1.0, 2.0)], -1)
    return np.clip(base + rs.randn(h, w, 3) * sig, 0, 255).astype(np.uint8)
