"""Host mirror of the reference's ``frame_analysis.py`` over the HIP forensic kernels.

`FrameForensicAnalyzer` keeps the reference class's constructor, methods, result dicts and
visible attributes (reference frame_analysis.py:22-395); each call is one `dfd_forensics`
through the C ABI, which resizes the frame to 256x256 on the GPU, runs the six signal kernels
on the library's stream and applies the reference's thresholds.  The temporal state lives in the
library per stream id; the attributes below read it back.
"""
from __future__ import annotations

import itertools
from typing import Optional

import numpy as np

from ._lib import Handle

_stream_ids = itertools.count(1)


class FrameForensicAnalyzer:
    def __init__(self, analysis_size=(256, 256), *, handle: Optional[Handle] = None, stream_id: Optional[int] = None):
        if tuple(analysis_size) != (256, 256):
            raise ValueError("the HIP forensic kernels are built for analysis_size=(256, 256) "
                             "(the only size the reference ever constructs, deepfake_detection.py:327)")
        self.analysis_size = tuple(analysis_size)
        self._handle = handle
        self.stream_id = next(_stream_ids) if stream_id is None else int(stream_id)
        self.weights = {'frequency': 0.25, 'noise': 0.20, 'ela': 0.20, 'edge': 0.15, 'color': 0.10,
                        'temporal': 0.10}                       # reference :49-56 (reported; applied in the library)
        self.last_stats = {}

    # ---- device
    @property
    def handle(self) -> Handle:
        if self._handle is None:
            from . import runtime

            self._handle = runtime.default_handle()
        return self._handle

    def _existing_handle(self) -> Optional[Handle]:
        """The handle if one exists already - state queries and reset must not need a GPU when
        nothing has run yet (there is then no state)."""
        if self._handle is None:
            from . import runtime

            return runtime.peek_default_handle()
        return self._handle

    def _state(self):
        h = self._existing_handle()
        return h.forensics_state(self.stream_id) if h is not None else (0, 0, False)

    # ---- reference attributes, read back from the library
    @property
    def frame_count(self) -> int:
        return self._state()[0]

    @property
    def temporal_diffs(self):
        """len() is what callers use (reference :376); values stay on the library side."""
        return range(self._state()[1])

    @property
    def prev_frame_gray(self):
        return True if self._state()[2] else None

    # ---- reference methods
    def _run(self, frame, full: bool, kind: str):
        frame = np.asarray(frame)
        if frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8 or frame.shape[0] < 1 or frame.shape[1] < 1:
            raise ValueError(f"expected a BGR uint8 image, got {frame.dtype} {frame.shape}")
        scores, prob, stats = self.handle.forensics(frame, full=full, stream_id=self.stream_id)
        self.last_stats = stats
        return {'scores': scores, 'fake_probability': prob, 'analysis_type': kind,
                'frame_number': int(stats['frame_count'])}

    def analyze(self, frame):
        """All six signals (reference :58-101)."""
        return self._run(frame, True, 'frame_forensic')

    def analyze_fast(self, frame):
        """frequency + temporal + edge only (reference :103-126)."""
        return self._run(frame, False, 'frame_forensic_fast')

    def reset(self):
        """reference :391-395"""
        h = self._existing_handle()
        if h is not None:
            h.forensics_reset(self.stream_id)
        self.last_stats = {}
