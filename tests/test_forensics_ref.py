"""CPU: the forensic oracle against the properties the reference's tests pin
(tests/test_algorithm.py:169-205, tests/test_functional.py:164-216, tests/test_reliability.py:134-147)."""
import numpy as np

import frames as F
from oracle.forensics_ref import ForensicsRef


def test_keys_ranges_and_weighted_sum():
    a = ForensicsRef()
    r = a.analyze(F.face_frame())
    assert set(r["scores"]) == {"frequency", "noise", "ela", "edge", "color", "temporal"}
    assert all(0.0 <= v <= 1.0 for v in r["scores"].values()) and 0.0 <= r["fake_probability"] <= 1.0
    manual = float(np.clip(sum(r["scores"][k] * a.weights[k] for k in a.weights), 0.0, 1.0))
    assert abs(r["fake_probability"] - manual) < 1e-6
    rf = a.analyze_fast(F.face_frame())
    assert set(rf["scores"]) == {"frequency", "temporal", "edge"} and a.frame_count == 2
    a.reset()
    assert a.frame_count == 0 and a.prev_frame_gray is None


def test_ordering_properties():
    smooth, noisy, edgy = F.smooth_image(), F.noisy_image(), F.gradient_image()
    rs, rn, re = ForensicsRef().analyze(smooth), ForensicsRef().analyze(noisy), ForensicsRef().analyze(edgy)
    assert rs["scores"]["frequency"] >= rn["scores"]["frequency"]
    assert ForensicsRef().analyze(np.full((256, 256, 3), 100, np.uint8))["scores"]["color"] >= rn["scores"]["color"]
    assert rs["scores"]["edge"] >= re["scores"]["edge"]


def test_determinism_frame():
    f = F.determinism_frame()
    r1, r2 = ForensicsRef().analyze(f), ForensicsRef().analyze(f)
    assert all(abs(r1["scores"][k] - r2["scores"][k]) < 1e-6 for k in r1["scores"])


def test_temporal_needs_five_diffs_and_ten_frames():
    a = ForensicsRef()
    f = F.face_frame()
    out = [a.analyze_fast(f)["scores"]["temporal"] for _ in range(12)]
    assert out[:5] == [0.0] * 5              # first frame + fewer than 5 differences
    assert out[5] == 0.0 and out[10] == 0.3  # frozen content counts only once frame_count > 10
