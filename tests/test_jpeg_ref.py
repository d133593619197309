"""CPU: pins the ELA codec oracle.  oracle/jpeg_ref.roundtrip_np (numpy restatement of libjpeg's
integer q90 4:2:0 encode->decode) must equal Pillow's libjpeg round trip byte for byte; the HIP
kernels are then compared with roundtrip_np's consumers in tests/test_forensics_gpu.py."""
import numpy as np
import pytest

import frames as F
from oracle import jpeg_ref as J


@pytest.mark.parametrize("name,img", [
    ("determinism", F.determinism_frame()),
    ("noisy", F.noisy_image()),
    ("gradient", F.gradient_image()),
    ("flat", np.full((256, 256, 3), 128, np.uint8)),
    ("black", np.zeros((64, 64, 3), np.uint8)),
    ("white", np.full((32, 48, 3), 255, np.uint8)),
    ("small", np.random.RandomState(3).randint(0, 256, (16, 32, 3)).astype(np.uint8)),
    ("saturated", (np.random.RandomState(4).rand(64, 64, 3) > 0.5).astype(np.uint8) * 255),
])
def test_numpy_libjpeg_equals_pillow(name, img):
    a, b = J.roundtrip_np(img, 90), J.roundtrip_pil(img, 90)
    assert a.shape == b.shape == img.shape
    assert np.array_equal(a, b), f"{name}: {(a != b).sum()} bytes differ, max {np.abs(a.astype(int) - b).max()}"


@pytest.mark.parametrize("q", [50, 75, 95])
def test_other_qualities(q):
    img = F.noisy_image((64, 64), seed=q)
    assert np.array_equal(J.roundtrip_np(img, q), J.roundtrip_pil(img, q))


def test_quant_table_q90():
    t = J.quant_table(J._LUMA, 90)
    assert t[0, 0] == 3 and t[7, 7] == 20 and t.min() >= 1


def test_rejects_ragged_sizes():
    with pytest.raises(ValueError):
        J.roundtrip_np(np.zeros((20, 16, 3), np.uint8))
