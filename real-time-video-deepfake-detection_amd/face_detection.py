"""Host mirror of the reference's ``face_detection.py`` over the HIP SSD detector.

Same functions, arguments and return conventions as reference face_detection.py:37-188: the DNN
branch (`dfd_detect_faces`) when the handle carries SSD weights, else - and after a DNN failure - the
Haar cascade (`dfd_detect_faces_haar`, `haar.py`) when the handle carries one, else ``[]``, the neutral
value the reference itself ends with (face_detection.py:63-68).
"""
from __future__ import annotations

import logging
from typing import List, Optional, Tuple

import numpy as np

from . import runtime
from ._lib import DfdError, Handle

log = logging.getLogger(__name__)
Box = Tuple[int, int, int, int]


def detect_bounding_box(frame, confidence_threshold: float = 0.5, *, handle: Optional[Handle] = None) -> List[Box]:
    """[(x, y, w, h), ...] of the faces in a BGR frame, in descending-confidence order."""
    try:
        if frame is None or getattr(frame, "size", 0) == 0:
            return []
        if len(frame.shape) < 2 or frame.shape[0] < 30 or frame.shape[1] < 30:
            return []
        if frame.ndim != 3 or frame.shape[2] != 3:
            return []
        h = handle or runtime.default_handle()
        if not h.has_detector:                      # no SSD weights (runtime.py): the reference's Haar fallback, or []
            return _detect_haar(frame, h) if h.has_haar else []
        try:
            return h.detect_faces(frame, confidence_threshold)
        except DfdError as e:                       # reference :58-66: DNN failure -> Haar retry
            log.warning("DNN face detection failed (%s); trying the Haar cascade", e)
            return _detect_haar(frame, h) if h.has_haar else []
    except (DfdError, ValueError, TypeError) as e:
        log.warning("face detection failed: %s", e)
        return []


def _detect_haar(frame, handle: Handle) -> List[Box]:
    """reference :108-123: detectMultiScale(gray, scaleFactor=1.1, minNeighbors=5, minSize=(30, 30))"""
    from . import haar

    return haar.detect_faces_haar(frame, handle)


def draw_bounding_boxes(frame, faces, color=(0, 255, 0), thickness: int = 2):
    """Copy of `frame` with rectangle outlines (reference :125-143; plain numpy, no cv2)."""
    out = np.array(frame, copy=True)
    H, W = out.shape[:2]
    t = max(1, int(thickness))
    for (x, y, w, h) in faces:
        x0, y0, x1, y1 = max(0, x), max(0, y), min(W, x + w), min(H, y + h)
        if x1 <= x0 or y1 <= y0:
            continue
        out[y0:min(y0 + t, y1), x0:x1] = color
        out[max(y1 - t, y0):y1, x0:x1] = color
        out[y0:y1, x0:min(x0 + t, x1)] = color
        out[y0:y1, max(x1 - t, x0):x1] = color
    return out


def extract_face_region(frame, face_box: Box, padding: int = 0):
    """View of the (padded, clamped) box (reference :145-168)."""
    x, y, w, h = face_box
    x0, y0 = max(0, x - padding), max(0, y - padding)
    x1, y1 = min(frame.shape[1], x + w + padding), min(frame.shape[0], y + h + padding)
    return frame[y0:y1, x0:x1]


def detect_and_extract_faces(frame, padding: int = 0):
    """[(face_region, (x, y, w, h)), ...] (reference :170-188)."""
    return [(extract_face_region(frame, b, padding), b) for b in detect_bounding_box(frame)]
