"""ORACLE (test infrastructure only): restatement of the reference's FrameForensicAnalyzer
(reference frame_analysis.py:22-395) on numpy, with cv2 calls replaced by oracle/imgproc_ref.py
(OpenCV algorithms restated; unverified against a real cv2 -> "parity unpinned") and the JPEG
round trip by oracle/jpeg_ref.py (pinned against Pillow's libjpeg).

Besides the six scores, every method also returns the raw statistics the thresholds act on
(`stats`), because the scores are step functions: parity tests compare the statistics within a
tolerance and the scores exactly on inputs away from the thresholds.

PINNED by the reference's tests only as far as they go: key set, [0,1] ranges, determinism
(tests/test_reliability.py:134-147), weighted-sum identity (tests/test_algorithm.py:199-205),
ordering properties (tests/test_algorithm.py:169-197), reset (tests/test_functional.py:205-216).
"""
from __future__ import annotations

from collections import deque

import numpy as np

from . import imgproc_ref as I
from . import jpeg_ref as J

WEIGHTS = {'frequency': 0.25, 'noise': 0.20, 'ela': 0.20, 'edge': 0.15, 'color': 0.10, 'temporal': 0.10}   # :49-56
FAST_WEIGHTS = {'frequency': 0.45, 'temporal': 0.25, 'edge': 0.30}                                           # :118


class ForensicsRef:
    def __init__(self, analysis_size=(256, 256)):
        self.analysis_size = analysis_size
        self.prev_frame_gray = None
        self.temporal_diffs = deque(maxlen=30)
        self.frame_count = 0
        h, w = analysis_size
        cy, cx = h // 2, w // 2
        yg, xg = np.ogrid[:h, :w]
        self._dist = np.sqrt((xg - cx) ** 2 + (yg - cy) ** 2)          # :40-46
        self._inner, self._mid, self._outer = min(h, w) // 8, min(h, w) // 4, min(h, w) // 2
        self.weights = dict(WEIGHTS)
        self.stats = {}

    # ------------------------------------------------------------------ drivers (:58-126)
    def _resize(self, frame):
        return I.resize_linear_u8(frame, self.analysis_size[0], self.analysis_size[1])

    def analyze(self, frame):
        self.frame_count += 1
        r = self._resize(frame)
        self.stats = {}
        scores = {'frequency': self.frequency(r), 'noise': self.noise(r), 'ela': self.ela(r),
                  'edge': self.edges(r), 'color': self.color(r), 'temporal': self.temporal(r)}
        combined = sum(scores[k] * self.weights[k] for k in self.weights)
        return {'scores': scores, 'fake_probability': float(np.clip(combined, 0.0, 1.0)),
                'analysis_type': 'frame_forensic', 'frame_number': self.frame_count}

    def analyze_fast(self, frame):
        self.frame_count += 1
        r = self._resize(frame)
        self.stats = {}
        scores = {'frequency': self.frequency(r), 'temporal': self.temporal(r), 'edge': self.edges(r)}
        combined = sum(scores[k] * FAST_WEIGHTS[k] for k in FAST_WEIGHTS)
        return {'scores': scores, 'fake_probability': float(np.clip(combined, 0.0, 1.0)),
                'analysis_type': 'frame_forensic_fast', 'frame_number': self.frame_count}

    # ------------------------------------------------------------------ signals
    def frequency(self, frame):                                         # :128-180
        gray = I.bgr2gray_u8(frame).astype(np.float32)
        mag = np.log1p(np.abs(np.fft.fftshift(np.fft.fft2(gray))))
        d = self._dist
        low, mid, high = d <= self._inner, (d > self._inner) & (d <= self._mid), (d > self._mid) & (d <= self._outer)
        lo, mi, hi = mag[low].mean(), mag[mid].mean(), mag[high].mean()
        total = lo + mi + hi + 1e-10
        hr, mr = hi / total, mi / total
        score = 0.0
        if hr < 0.18:
            score += 0.4
        elif hr < 0.22:
            score += 0.2
        mv = mag[mid]
        mid_cv = np.std(mv) / (np.mean(mv) + 1e-10)
        if mid_cv > 0.6:
            score += 0.25
        elif mid_cv > 0.45:
            score += 0.1
        if mr > 0.45 and hr < 0.2:
            score += 0.15
        self.stats.update(freq_low=float(lo), freq_mid=float(mi), freq_high=float(hi), freq_high_ratio=float(hr),
                          freq_mid_ratio=float(mr), freq_mid_cv=float(mid_cv))
        return float(np.clip(score, 0.0, 1.0))

    @staticmethod
    def _blocks(a, size=32):
        h, w = a.shape
        return [a[i:i + size, j:j + size] for i in range(0, h - size + 1, size) for j in range(0, w - size + 1, size)]

    def noise(self, frame):                                             # :182-225
        gray = I.bgr2gray_u8(frame).astype(np.float32)
        resid = gray - I.gaussian5_f32(gray)
        stds = np.array([np.std(b) for b in self._blocks(resid)])
        if len(stds) < 4:
            return 0.0
        mean_noise = np.mean(stds)
        cv = np.std(stds) / (mean_noise + 1e-10)
        score = 0.0
        if cv > 0.7:
            score += 0.5
        elif cv > 0.5:
            score += 0.25
        if mean_noise < 1.0:
            score += 0.3
        elif mean_noise < 2.0:
            score += 0.1
        self.stats.update(noise_mean=float(mean_noise), noise_cv=float(cv))
        return float(np.clip(score, 0.0, 1.0))

    def ela(self, frame):                                               # :227-276
        recompressed = J.roundtrip_np(frame, 90)
        diff = np.abs(frame.astype(np.int16) - recompressed.astype(np.int16)).astype(np.uint8)
        dg = I.bgr2gray_u8(diff).astype(np.float32)
        means = np.array([np.mean(b) for b in self._blocks(dg)])
        if len(means) < 4:
            return 0.0
        ela_mean = np.mean(means)
        cv = np.std(means) / (ela_mean + 1e-10)
        score = 0.0
        if cv > 0.9:
            score += 0.5
        elif cv > 0.6:
            score += 0.2
        if ela_mean > 15:
            score += 0.2
        elif ela_mean > 10:
            score += 0.1
        self.stats.update(ela_mean=float(ela_mean), ela_cv=float(cv))
        return float(np.clip(score, 0.0, 1.0))

    def edges(self, frame):                                             # :278-309
        gray = I.bgr2gray_u8(frame)
        e = I.canny_u8(gray, 50, 150)
        density = np.sum(e > 0) / e.size
        lap_var = np.var(I.laplacian_i32(gray).astype(np.float64))
        score = 0.0
        if density < 0.02:
            score += 0.35
        elif density < 0.04:
            score += 0.15
        if lap_var < 50:
            score += 0.3
        elif lap_var < 100:
            score += 0.1
        self.stats.update(edge_density=float(density), lap_var=float(lap_var))
        return float(np.clip(score, 0.0, 1.0))

    def color(self, frame):                                             # :311-347
        hsv = I.bgr2hsv_u8(frame)
        sat_std = np.std(hsv[:, :, 1].astype(np.float32))
        val_std = np.std(hsv[:, :, 2].astype(np.float32))
        hues = len(np.unique(hsv[:, :, 0]))
        score = 0.0
        if sat_std < 15:
            score += 0.3
        elif sat_std < 25:
            score += 0.1
        if val_std < 15:
            score += 0.25
        elif val_std < 25:
            score += 0.1
        if hues < 30:
            score += 0.25
        elif hues < 50:
            score += 0.1
        self.stats.update(sat_std=float(sat_std), val_std=float(val_std), unique_hues=float(hues))
        return float(np.clip(score, 0.0, 1.0))

    def temporal(self, frame):                                          # :349-389
        gray = I.bgr2gray_u8(frame).astype(np.float32)
        if self.prev_frame_gray is None:
            self.prev_frame_gray = gray
            self.stats.update(mean_diff=-1.0)
            return 0.0
        mean_diff = np.mean(np.abs(gray - self.prev_frame_gray))
        self.temporal_diffs.append(mean_diff)
        self.prev_frame_gray = gray
        self.stats.update(mean_diff=float(mean_diff))
        if len(self.temporal_diffs) < 5:
            return 0.0
        diffs = np.array(self.temporal_diffs)
        cv = np.std(diffs) / (np.mean(diffs) + 1e-10)
        score = 0.0
        if cv > 1.5:
            score += 0.4
        elif cv > 1.0:
            score += 0.2
        if mean_diff < 0.3 and self.frame_count > 10:
            score += 0.3
        elif mean_diff < 0.8 and self.frame_count > 10:
            score += 0.1
        self.stats.update(temporal_cv=float(cv))
        return float(np.clip(score, 0.0, 1.0))

    def reset(self):                                                    # :391-395
        self.prev_frame_gray = None
        self.temporal_diffs.clear()
        self.frame_count = 0
