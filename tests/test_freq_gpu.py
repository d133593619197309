"""GPU: compute_frequency_features (reference model.py:105-149) vs the oracle, and the properties the
reference tests pin (tests/test_algorithm.py:212-244, tests/test_reliability.py:149-155)."""
import numpy as np
import pytest

import frames as F
from oracle import imgproc_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("img", [np.random.RandomState(123).randint(0, 255, (200, 200, 3)).astype(np.uint8),
                                 np.random.RandomState(5).randint(0, 255, (300, 300, 3)).astype(np.uint8),
                                 F.face_frame(), F.gradient_image(), F.natural_like(224, 224, 9)],
                         ids=["rand200", "rand300", "face", "gradient", "natural224"])
def test_matches_oracle(pkg, b0_handle, img):
    got = pkg.model.compute_frequency_features(img, size=224, handle=b0_handle)
    want = R.compute_frequency_features(img, 224)
    assert got.shape == (2, 224, 224) and got.dtype == np.float32
    assert got.min() >= -0.01 and got.max() <= 1.01
    assert np.abs(got - want).max() <= 2e-3, [float(np.abs(got[c] - want[c]).max()) for c in (0, 1)]
    assert np.abs(got - want).mean() <= 1e-4


def test_reference_properties(pkg, b0_handle):
    z = pkg.model.compute_frequency_features(np.zeros((200, 200, 3), np.uint8), handle=b0_handle)
    r = pkg.model.compute_frequency_features(np.random.RandomState(1).randint(0, 255, (200, 200, 3)).astype(np.uint8), handle=b0_handle)
    assert np.all(z == 0.0) and not np.allclose(z, r)                    # flat image: both channels degenerate to zeros
    img = np.random.RandomState(123).randint(0, 255, (200, 200, 3)).astype(np.uint8)
    a = pkg.model.compute_frequency_features(img, handle=b0_handle)
    b = pkg.model.compute_frequency_features(img, handle=b0_handle)
    assert np.array_equal(a, b)                                           # deterministic
    g = pkg.model.compute_frequency_features(R.bgr2gray_u8(img), handle=b0_handle)   # 2-D gray input
    assert np.array_equal(g, a)
    with pytest.raises(ValueError):
        pkg.model.compute_frequency_features(img, size=128, handle=b0_handle)
