"""CPU: TemporalTracker (product host logic) and its oracle against the reference's
known-answer tests (tests/golden/tracker_kats.json) and against each other on random
streams.  Votes, counts and verdict strings must match exactly; float statistics exactly too
(both sides do the same f64 arithmetic)."""
import json
import os

import numpy as np
import pytest

from oracle.tracker_ref import TrackerRef

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "tracker_kats.json")))["cases"]


def _drive(t, seg, is_ref):
    if seg.get("reset"):
        t.reset()
    if "alternate" in seg:
        a, b, n = seg["alternate"]
        for i in range(n):
            t.update(a if i % 2 == 0 else b)
    for p, n in seg["runs"]:
        for _ in range(n):
            t.update(p)


def _view(t, is_ref):
    if is_ref:
        return dict(level=t.confidence_level(), stats=t.voting_stats(), avg=t.temporal_average(),
                    stab=t.stability(), votes_len=len(t.votes), scores_len=len(t.scores))
    return dict(level=t.get_confidence_level(), stats=t.get_voting_stats(), avg=t.get_temporal_average(),
                stab=t.get_stability_score(), votes_len=len(t.frame_classifications),
                scores_len=len(t.score_history))


@pytest.mark.parametrize("case", KATS, ids=[c["id"] for c in KATS])
@pytest.mark.parametrize("which", ["product", "oracle"])
def test_reference_known_answers(pkg, case, which):
    t = TrackerRef(**case["init"]) if which == "oracle" else pkg.tracker.TemporalTracker(**case["init"])
    for seg in case["segments"]:
        _drive(t, seg, which == "oracle")
        v = _view(t, which == "oracle")
        e = seg["expect"]
        if "level" in e:
            assert v["level"] == e["level"]
        for k in ("fake_count", "real_count", "total_frames"):
            if k in e:
                assert v["stats"][k] == e[k]
        for k in ("votes_len", "scores_len"):
            if k in e:
                assert v[k] == e[k]
        if "average" in e:
            assert abs(v["avg"] - e["average"]) < e["average_tol"]
        if "stability_gt" in e:
            assert v["stab"] > e["stability_gt"]
        if "stability_lt" in e:
            assert v["stab"] < e["stability_lt"]
        if "stability_eq" in e:
            assert v["stab"] == e["stability_eq"]


@pytest.mark.parametrize("seed,thr,win", [(0, 0.5, 10), (1, 0.55, 10), (2, 0.75, 5), (3, 0.5, 7)])
def test_product_equals_oracle_on_random_streams(pkg, seed, thr, win):
    rs = np.random.RandomState(seed)
    a = pkg.tracker.TemporalTracker(window_size=60, voting_window=win, detection_threshold=thr)
    b = TrackerRef(window_size=60, voting_window=win, detection_threshold=thr)
    probs = rs.rand(400)
    probs[rs.rand(400) < 0.05] = thr           # exact-threshold hits must vote REAL
    for i, p in enumerate(probs):
        p = None if i % 37 == 36 else float(p)
        a.update(p)
        b.update(p)
        assert a.get_confidence_level() == b.confidence_level()
        assert a.get_voting_stats() == b.voting_stats()
        assert a.get_temporal_average() == b.temporal_average()
        assert a.get_stability_score() == b.stability()
        assert a.get_weighted_average() == b.weighted_average()
        assert a.detect_anomalies() == b.anomalies()
        if i == 250:
            a.reset()
            b.reset()


def test_replay_equals_sequential(pkg):
    rs = np.random.RandomState(5)
    probs = rs.rand(64).astype(np.float32)
    a = pkg.tracker.TemporalTracker()
    b = pkg.tracker.TemporalTracker()
    for p in probs:
        a.update(float(p))
    for i in range(0, 64, 8):                   # waves of 8 gathered frames
        b.replay(probs[i:i + 8])
    assert a.get_voting_stats() == b.get_voting_stats()
    assert a.get_confidence_level() == b.get_confidence_level()
    assert list(a.score_history) == list(b.score_history)


def test_trigger_needs_history_average_stability_and_cooldown(pkg):
    t = pkg.tracker.TemporalTracker(window_size=60, high_confidence_threshold=0.6)
    for _ in range(29):
        t.update(0.9)
    assert not t.should_trigger_forensic_analysis()          # < window_size // 2 scores
    t.update(0.9)
    assert t.should_trigger_forensic_analysis()
    assert not t.should_trigger_forensic_analysis()          # 5 s cooldown
    r = TrackerRef()
    for _ in range(30):
        r.update(0.9)
    assert r.should_trigger(now=100.0) and not r.should_trigger(now=102.0) and r.should_trigger(now=106.0)
