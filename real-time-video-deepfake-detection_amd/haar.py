"""Haar cascade fallback of the reference's face detection (reference face_detection.py:12,108-123).

The reference builds ``cv2.CascadeClassifier(cv2.data.haarcascades + 'haarcascade_frontalface_default.xml')`` at
import and uses it whenever the SSD model files are missing - which is how the reference runs as shipped.
This module reads that XML format (OpenCV's "new" cascade layout: BOOST stages of stumps over upright HAAR features)
into the arrays `weights.pack_all(..., haar=...)` puts into the blob; csrc/haar_api.hip evaluates them.

The XML itself ships with OpenCV, not with the reference tree or this one: point ``$DFD_HAAR_CASCADE`` at it.
Not supported (rejected with a clear error): tilted features, trees deeper than stumps, LBP / HOG cascades, the old
(OpenCV 1.x) cascade layout.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Dict

import numpy as np


def load_cascade_xml(path_or_text: str) -> Dict[str, np.ndarray]:
    text = path_or_text if path_or_text.lstrip().startswith("<") else open(path_or_text).read()
    root = ET.fromstring(text)
    cas = root.find("cascade")
    if cas is None:
        raise ValueError("Haar XML: no <cascade> element (old-format cascades are not supported)")
    if (cas.findtext("stageType") or "").strip() != "BOOST" or (cas.findtext("featureType") or "").strip() != "HAAR":
        raise ValueError("Haar XML: only BOOST cascades of HAAR features are supported")
    win = (int(cas.findtext("width")), int(cas.findtext("height")))
    stages, stumps = [], []
    for st in cas.find("stages"):
        first = len(stumps)
        for wk in st.find("weakClassifiers"):
            nodes = [float(v) for v in wk.findtext("internalNodes").split()]
            leaves = [float(v) for v in wk.findtext("leafValues").split()]
            if len(nodes) != 4 or len(leaves) != 2 or nodes[0] > 0 or nodes[1] > 0:
                raise ValueError("Haar XML: only stump weak classifiers (one internal node) are supported")
            # internalNodes: left right featureIdx threshold; left / right <= 0 are leaf indices (negated)
            stumps.append((nodes[2], nodes[3], leaves[int(-nodes[0])], leaves[int(-nodes[1])]))
        stages.append((first, len(stumps) - first, float(st.findtext("stageThreshold"))))
    rects = []
    for f in cas.find("features"):
        if int((f.findtext("tilted") or "0").strip()) != 0:
            raise ValueError("Haar XML: tilted features are not supported")
        rs = [[float(v) for v in r.text.split()] for r in f.find("rects")]
        if not 2 <= len(rs) <= 3:
            raise ValueError("Haar XML: a feature has 2 or 3 rectangles")
        while len(rs) < 3:
            rs.append([0, 0, 0, 0, 0.0])
        rects.append(rs)
    return {"haar.win": np.asarray(win, np.float32), "haar.stages": np.asarray(stages, np.float32),
            "haar.stumps": np.asarray(stumps, np.float32), "haar.rects": np.asarray(rects, np.float32).reshape(len(rects), 15)}


def detect_faces_haar(frame, handle):
    """reference face_detection.py:108-123: [(x, y, w, h), ...] with scaleFactor 1.1, minNeighbors 5, minSize 30"""
    return handle.detect_faces_haar(frame, 1.1, 5, 30)
