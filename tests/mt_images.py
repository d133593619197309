"""Seeded face-crop-like images for the MTCNN tests (sizes chosen to cover 4-8 pyramid levels, a crop below
the minimum face size, and both 'face found' and 'no face' outcomes with weights.seeded_mtcnn_state_dict(0))."""
import numpy as np


def textured(h, w, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 110 + 60 * np.sin(xx / 17.0)[..., None] * np.cos(yy / 23.0)[..., None]
    return np.clip(base + rs.randn(h, w, 3) * 25 + np.array([10, -5, 20]), 0, 255).astype(np.uint8)


CASES = [(300, 280, 3), (180, 200, 4), (90, 75, 5), (40, 33, 6), (161, 240, 7), (12, 40, 8)]


def images():
    return [textured(h, w, s) for h, w, s in CASES]
