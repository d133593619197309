"""TemporalTracker: the 10-frame majority vote and the running statistics around it.

Host logic with the constructor, attributes and methods of the reference class
(reference deepfake_detection.py:93-290).  Votes and verdicts are integer work and are
bit-exact; the vote counts are maintained incrementally (O(1) per frame) rather than by
re-walking the queue, and the hot path prints nothing (the reference prints a debug line per
frame at :138).  `replay` applies a whole wave of gathered probabilities in frame order: that
is what every rank calls after the multi-GPU all-gather so that all ranks hold identical
vote state.
"""
from __future__ import annotations

import time
from collections import deque
from typing import Iterable, Optional

import numpy as np

FAKE = "FAKE"
REAL = "REAL"
UNCERTAIN = "UNCERTAIN"


class TemporalTracker:
    def __init__(self, window_size: int = 60, high_confidence_threshold: float = 0.6,
                 voting_window: int = 10, detection_threshold: float = 0.5):
        self.window_size = window_size
        self.high_confidence_threshold = high_confidence_threshold
        self.voting_window = voting_window
        self.detection_threshold = detection_threshold
        self.score_history = deque(maxlen=window_size)
        self.variance_history = deque(maxlen=30)
        self.last_alert_time = 0
        self.alert_cooldown = 5                      # seconds
        self.frame_classifications = deque(maxlen=voting_window)
        self.current_verdict: Optional[str] = None
        self._fake_votes = 0                         # FAKE entries currently in the vote queue

    # ------------------------------------------------------------------ update
    def update(self, fake_probability) -> None:
        """Push one frame's probability; ``None`` is ignored (reference :123-124)."""
        if fake_probability is None:
            return
        self.score_history.append(fake_probability)
        if len(self.score_history) >= 5:
            n = len(self.score_history)
            recent = [self.score_history[i] for i in range(n - 5, n)]
            self.variance_history.append(np.var(recent))
        vote = FAKE if fake_probability > self.detection_threshold else REAL    # strict '>'
        q = self.frame_classifications
        if len(q) == q.maxlen and q[0] == FAKE:
            self._fake_votes -= 1                    # the entry about to fall off the queue
        q.append(vote)
        if vote == FAKE:
            self._fake_votes += 1
        self._update_verdict()

    def replay(self, probabilities: Iterable) -> None:
        """Apply probabilities in the given (frame) order."""
        for p in probabilities:
            self.update(None if p is None else float(p))

    def _update_verdict(self) -> None:
        total = len(self.frame_classifications)
        if total < self.voting_window:               # window not full yet -> stay undecided
            self.current_verdict = None
            return
        real = total - self._fake_votes
        self.current_verdict = FAKE if self._fake_votes > real else REAL       # tie -> REAL

    # ------------------------------------------------------------------ statistics
    def get_temporal_average(self) -> float:
        if not self.score_history:
            return 0.0
        return sum(self.score_history) / len(self.score_history)

    def get_weighted_average(self) -> float:
        if not self.score_history:
            return 0.0
        scores = list(self.score_history)
        w = np.linspace(0.5, 1.0, len(scores))
        return sum(s * x for s, x in zip(scores, w)) / sum(w)

    def get_stability_score(self) -> float:
        if len(self.score_history) < 10:
            return 0.0
        scores = list(self.score_history)
        mean = sum(scores) / len(scores)
        variance = sum((x - mean) ** 2 for x in scores) / len(scores)
        return 1.0 - min(variance * 4, 1.0)

    def detect_anomalies(self) -> float:
        if len(self.variance_history) < 10:
            return 0.0
        return min(np.mean(list(self.variance_history)) * 10, 1.0)

    def should_trigger_forensic_analysis(self) -> bool:
        if len(self.score_history) < self.window_size // 2:
            return False
        now = time.time()
        if (self.get_temporal_average() > self.high_confidence_threshold
                and self.get_stability_score() > 0.7
                and now - self.last_alert_time > self.alert_cooldown):
            self.last_alert_time = now
            return True
        return False

    def get_confidence_level(self) -> str:
        return UNCERTAIN if self.current_verdict is None else self.current_verdict

    def get_voting_stats(self) -> dict:
        total = len(self.frame_classifications)
        return {"fake_count": self._fake_votes, "real_count": total - self._fake_votes,
                "total_frames": total}

    def reset(self) -> None:
        self.score_history.clear()
        self.variance_history.clear()
        self.last_alert_time = 0
        self.frame_classifications.clear()
        self._fake_votes = 0
        self.current_verdict = None
