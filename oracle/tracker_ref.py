"""ORACLE (test infrastructure only): plain restatement of the reference's TemporalTracker.

Follows reference deepfake_detection.py:93-290 line by line in behaviour, with lists instead
of deques and every statistic recomputed from scratch on each query, so that the product's
incremental implementation (package deepfake_detection.TemporalTracker) has something
independent to be compared with.  PINNED by the reference's own known-answer tests:
tests/test_functional.py:223-305, tests/test_algorithm.py:50-154,251-278,
tests/test_reliability.py:309-320 (transcribed as data in tests/golden/tracker_kats.json).
"""
from __future__ import annotations

import time

import numpy as np


class TrackerRef:
    def __init__(self, window_size=60, high_confidence_threshold=0.6, voting_window=10,
                 detection_threshold=0.5):
        self.window_size = window_size
        self.high_confidence_threshold = high_confidence_threshold
        self.voting_window = voting_window
        self.detection_threshold = detection_threshold
        self.scores = []          # last `window_size` probabilities         (:111)
        self.variances = []       # last 30 five-sample variances            (:112)
        self.votes = []           # last `voting_window` 'FAKE'/'REAL'       (:117)
        self.verdict = None       #                                          (:118)
        self.last_alert_time = 0
        self.alert_cooldown = 5

    def update(self, p):
        if p is None:                                   # :123-124
            return
        self.scores = (self.scores + [p])[-self.window_size:]
        if len(self.scores) >= 5:                       # :129-132
            self.variances = (self.variances + [np.var(self.scores[-5:])])[-30:]
        vote = 'FAKE' if p > self.detection_threshold else 'REAL'     # strict '>'  (:135)
        self.votes = (self.votes + [vote])[-self.voting_window:]
        # _update_verdict (:146-196)
        if len(self.votes) < self.voting_window:
            self.verdict = None
            return
        fake = sum(1 for v in self.votes if v == 'FAKE')
        real = len(self.votes) - fake
        self.verdict = 'FAKE' if fake > real else 'REAL'              # tie -> REAL (:175-178)

    def confidence_level(self):                         # :252-258
        return 'UNCERTAIN' if self.verdict is None else self.verdict

    def temporal_average(self):                         # :198-202
        return 0.0 if not self.scores else sum(self.scores) / len(self.scores)

    def weighted_average(self):                         # :204-212
        if not self.scores:
            return 0.0
        w = np.linspace(0.5, 1.0, len(self.scores))
        return sum(s * x for s, x in zip(self.scores, w)) / sum(w)

    def stability(self):                                # :214-221
        if len(self.scores) < 10:
            return 0.0
        m = sum(self.scores) / len(self.scores)
        var = sum((x - m) ** 2 for x in self.scores) / len(self.scores)
        return 1.0 - min(var * 4, 1.0)

    def anomalies(self):                                # :223-233
        if len(self.variances) < 10:
            return 0.0
        return min(np.mean(self.variances) * 10, 1.0)

    def should_trigger(self, now=None):                 # :235-250
        if len(self.scores) < self.window_size // 2:
            return False
        now = time.time() if now is None else now
        if (self.temporal_average() > self.high_confidence_threshold and self.stability() > 0.7
                and now - self.last_alert_time > self.alert_cooldown):
            self.last_alert_time = now
            return True
        return False

    def voting_stats(self):                             # :260-268
        fake = sum(1 for v in self.votes if v == 'FAKE')
        return {'fake_count': fake, 'real_count': len(self.votes) - fake, 'total_frames': len(self.votes)}

    def reset(self):                                    # :270-283
        self.scores, self.variances, self.votes = [], [], []
        self.verdict = None
        self.last_alert_time = 0
