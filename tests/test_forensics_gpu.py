"""GPU parity for the six forensic signals through the C ABI vs oracle/forensics_ref.py.

Scores are step functions of statistics, so: (1) statistics must agree within STAT_RTOL
(integer-derived ones - ELA block means, edge density, hue count, frame difference - exactly);
(2) scores and the weighted probability must be identical whenever no statistic sits within
the tolerance of one of the reference's thresholds (the fixtures are chosen that way; the test
asserts it instead of assuming it)."""
import numpy as np
import pytest

import frames as F
from oracle.forensics_ref import ForensicsRef

pytestmark = pytest.mark.gpu

STAT_RTOL = 2e-4
EXACT = ("ela_mean", "edge_density", "unique_hues", "mean_diff")
THRESHOLDS = {"freq_high_ratio": (0.18, 0.2, 0.22), "freq_mid_cv": (0.45, 0.6), "freq_mid_ratio": (0.45,),
              "noise_cv": (0.5, 0.7), "noise_mean": (1.0, 2.0), "ela_cv": (0.6, 0.9), "ela_mean": (10, 15),
              "edge_density": (0.02, 0.04), "lap_var": (50, 100), "sat_std": (15, 25), "val_std": (15, 25),
              "unique_hues": (30, 50), "temporal_cv": (1.0, 1.5), "mean_diff": (0.3, 0.8)}


def _compare(got_scores, got_prob, got_stats, ref, res):
    knife = False
    for k, want in ref.stats.items():
        have = got_stats[k]
        if k in EXACT:
            assert have == want, (k, have, want)
        else:
            assert abs(have - want) <= STAT_RTOL * max(1.0, abs(want)), (k, have, want)
        for t in THRESHOLDS.get(k, ()):
            if abs(want - t) <= 2 * STAT_RTOL * max(1.0, abs(t)):
                knife = True
    assert set(got_scores) == set(res["scores"])
    if not knife:
        for k in res["scores"]:
            assert got_scores[k] == res["scores"][k], (k, got_scores[k], res["scores"][k], ref.stats)
        assert got_prob == res["fake_probability"]
    return knife


FRAMES = {"determinism": F.determinism_frame, "noisy": F.noisy_image, "gradient": F.gradient_image,
          "smooth": F.smooth_image, "face_vga": F.face_frame, "blank": F.blank_frame,
          "natural_720p": F.natural_like, "face_1080p": lambda: F.face_frame(1920, 1080, 5)}


@pytest.mark.parametrize("name", list(FRAMES))
def test_full_analysis_matches_oracle(b0_handle, name):
    frame = FRAMES[name]()
    ref = ForensicsRef()
    res = ref.analyze(frame)
    b0_handle.forensics_reset(900)
    scores, prob, stats = b0_handle.forensics(frame, full=True, stream_id=900)
    knife = _compare(scores, prob, stats, ref, res)
    assert not knife, f"fixture {name} sits on a threshold; pick another"
    assert set(scores) == {"frequency", "noise", "ela", "edge", "color", "temporal"}
    assert all(0.0 <= v <= 1.0 for v in scores.values()) and 0.0 <= prob <= 1.0


def test_stream_schedule_and_temporal_state(b0_handle):
    """14 frames with the reference's full/fast schedule (deepfake_detection.py:509-512) on one
    stream: temporal deque, frame counter and every per-frame result follow the oracle."""
    rs = np.random.RandomState(11)
    base = F.natural_like(480, 640, 8).astype(np.int16)
    ref = ForensicsRef()
    sid = 901
    b0_handle.forensics_reset(sid)
    for i in range(14):
        jitter = rs.randint(-3, 4, base.shape) if i not in (6, 7) else 0      # frames 6,7 repeat frame 5: zero diff
        if i not in (6, 7):
            cur = np.clip(base + jitter + i, 0, 255).astype(np.uint8)
        full = i % 3 == 0
        res = ref.analyze(cur) if full else ref.analyze_fast(cur)
        scores, prob, stats = b0_handle.forensics(cur, full=full, stream_id=sid)
        _compare(scores, prob, stats, ref, res)
        fc, nd, hp = b0_handle.forensics_state(sid)
        assert (fc, nd, hp) == (ref.frame_count, len(ref.temporal_diffs), ref.prev_frame_gray is not None)
    b0_handle.forensics_reset(sid)
    assert b0_handle.forensics_state(sid) == (0, 0, False)


def test_streams_are_independent_and_deterministic(b0_handle):
    f1, f2 = F.face_frame(seed=1), F.face_frame(seed=2)
    for sid in (910, 911):
        b0_handle.forensics_reset(sid)
    a1 = b0_handle.forensics(f1, True, 910)
    b0_handle.forensics(f2, True, 911)
    b0_handle.forensics_reset(910)
    a2 = b0_handle.forensics(f1, True, 910)
    # same frame, fresh state -> bit-identical (reference test_reliability.py:134-147); NaN marks "not computed"
    assert a1[0] == a2[0] and a1[1] == a2[1]
    assert all(a1[2][k] == a2[2][k] or (np.isnan(a1[2][k]) and np.isnan(a2[2][k])) for k in a1[2])
    assert b0_handle.forensics_state(911)[0] == 1


def test_analyzer_class_surface(pkg, b0_handle):
    """FrameForensicAnalyzer host mirror: the behaviours the reference tests pin
    (tests/test_functional.py:164-216, tests/test_algorithm.py:169-205)."""
    an = pkg.frame_analysis.FrameForensicAnalyzer(analysis_size=(256, 256), handle=b0_handle)
    frame = F.face_frame()
    r = an.analyze(frame)
    assert {"frequency", "noise", "ela", "edge", "color", "temporal"} <= set(r["scores"])
    assert r["analysis_type"] == "frame_forensic" and r["frame_number"] == 1
    manual = float(np.clip(sum(r["scores"][k] * an.weights[k] for k in an.weights), 0.0, 1.0))
    assert abs(r["fake_probability"] - manual) < 1e-6
    rf = an.analyze_fast(frame)
    assert set(rf["scores"]) == {"frequency", "temporal", "edge"} and rf["analysis_type"] == "frame_forensic_fast"
    assert an.frame_count == 2 and an.prev_frame_gray is not None
    an.reset()
    assert an.frame_count == 0 and an.prev_frame_gray is None and len(an.temporal_diffs) == 0
    smooth, noisy, edgy = F.smooth_image(), F.noisy_image(), F.gradient_image()
    rs_, rn = an.analyze(smooth), (an.reset(), an.analyze(noisy))[1]
    assert rs_["scores"]["frequency"] >= rn["scores"]["frequency"]
    an.reset()
    uni = an.analyze(np.full((256, 256, 3), 100, np.uint8))
    assert uni["scores"]["color"] >= rn["scores"]["color"]
    an.reset()
    assert rs_["scores"]["edge"] >= an.analyze(edgy)["scores"]["edge"]
    with pytest.raises(ValueError):
        pkg.frame_analysis.FrameForensicAnalyzer(analysis_size=(128, 128))


@pytest.mark.parametrize("w,h", [(160, 120), (320, 240), (500, 300), (1280, 720), (1920, 1080), (1, 1)])
def test_any_resolution(b0_handle, w, h):
    frame = np.random.RandomState(w + h).randint(0, 255, (h, w, 3)).astype(np.uint8)
    b0_handle.forensics_reset(920)
    scores, prob, _ = b0_handle.forensics(frame, True, 920)
    assert 0.0 <= prob <= 1.0
