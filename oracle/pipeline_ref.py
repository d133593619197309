"""ORACLE (test infrastructure only): the two per-frame orchestrations of the reference,
composed from the other oracle modules.

  PredictRef.predict   reference deepfake_detection.py:588-686 (`DeepfakeDetector.predict`):
                       frame_count += 1 FIRST, forensics (full when count % 3 == 0), detect, ALL
                       faces analysed and voted one by one, frame-forensic vote when no face.
  PredictRef.request   reference backend_server.py:147-233 (`/analyze` body): forensics BEFORE the
                       counter moves, faces[0] only, one vote per request.

analyze_face = CLAHE -> MTCNN.forward when `mtcnn_sd` is given (mtcnn_ref; a crop without a face yields no
prediction: skipped by predict, frame-only response from the server) -> bilinear 224 -> normalise -> B0 -> sigmoid
-> calibration (identity: no calibrator.pkl) -> +0.10 if h<80 or w<80 -> clip
(reference deepfake_detection.py:357-406,445-455,489-550).  PARITY UNPINNED as a whole (it inherits
that status from b0_ref / imgproc_ref / ssd_ref); the vote logic inside it is pinned (tracker_ref).
"""
from __future__ import annotations

import numpy as np
import torch

from . import b0_ref, haar_ref, imgproc_ref, mtcnn_ref, ssd_ref
from .forensics_ref import ForensicsRef
from .tracker_ref import TrackerRef


class PredictRef:
    def __init__(self, b0_sd, ssd_sd, ssd_arch, detection_threshold=0.5, mtcnn_sd=None, haar_cascade=None):
        self.b0_sd, self.ssd_sd, self.arch, self.mtcnn_sd = b0_sd, ssd_sd, ssd_arch, mtcnn_sd
        self.haar_cascade = haar_cascade
        self.tracker = TrackerRef(window_size=60, high_confidence_threshold=0.6, voting_window=10,
                                  detection_threshold=detection_threshold)
        self.analyzer = ForensicsRef()
        self.frame_count = 0
        self.full_forensic_interval = 3

    def detect(self, frame):
        """reference face_detection.py:37-68: the DNN when its model files were loaded, the Haar cascade otherwise
        (:58-61; the shipped configuration); [] for empty / tiny frames."""
        if self.ssd_sd is not None:
            return ssd_ref.detect_bounding_box(self.ssd_sd, self.arch, frame)
        if frame is None or frame.size == 0 or frame.ndim < 2 or frame.shape[0] < 30 or frame.shape[1] < 30:
            return []
        if self.haar_cascade is None:
            return []
        boxes, _ = haar_ref.detect(frame, self.haar_cascade)         # :108-123 (1.1, 5, 30x30)
        return [tuple(int(v) for v in b) for b in boxes]

    def forensics(self, frame):                                     # :504-515
        if self.frame_count % self.full_forensic_interval == 0:
            return self.analyzer.analyze(frame)
        return self.analyzer.analyze_fast(frame)

    def analyze_face(self, face):                                   # :517-550
        pre = imgproc_ref.preprocess_face_quality(face)
        if self.mtcnn_sd is not None:                               # :376-380
            aligned = mtcnn_ref.mtcnn_forward(self.mtcnn_sd, np.ascontiguousarray(pre[..., ::-1]))
            if aligned is None:
                return None, None
            pre = np.ascontiguousarray(aligned.transpose(1, 2, 0)[..., ::-1]).astype(np.uint8)
        x = torch.from_numpy(imgproc_ref.crop_resize_normalize(pre)).unsqueeze(0)
        logit = b0_ref.forward(self.b0_sd, x).squeeze()
        p = torch.sigmoid(logit).item()
        h, w = face.shape[:2]
        return float(np.clip(p + (0.10 if (h < 80 or w < 80) else 0.0), 0, 1)), float(logit)

    def predict(self, frame):
        self.frame_count += 1
        forensic = self.forensics(frame)
        faces = self.detect(frame)
        face_results, level = [], None          # None = the reference's local is unassigned (it raises at :679 when
        # faces were detected and every analyze_face returned None)
        if len(faces) > 0:
            for (x, y, w, h) in faces:
                p, logit = self.analyze_face(frame[y:y + h, x:x + w])
                if p is None:                                       # :616-617
                    continue
                self.tracker.update(p)
                level = self.tracker.confidence_level()
                face_results.append({'face_prob': p, 'logit': logit, 'bbox': {'x': x, 'y': y, 'w': w, 'h': h}})
        else:
            self.tracker.update(forensic['fake_probability'])
            level = self.tracker.confidence_level()
        return {'frame_count': self.frame_count, 'faces_detected': len(faces), 'face_results': face_results,
                'frame_forensic': forensic,
                'confidence_level': level if faces or self.frame_count > 1 else 'UNCERTAIN',
                'temporal_average': float(self.tracker.temporal_average()),
                'stability_score': float(self.tracker.stability()),
                'analysis_mode': 'face+frame' if len(faces) > 0 else 'frame_only',
                'votes': self.tracker.voting_stats()}

    def request(self, frame):
        forensic = self.forensics(frame)
        fprob = forensic['fake_probability']
        faces = self.detect(frame)
        self.frame_count += 1
        if len(faces) > 0:
            x, y, w, h = faces[0]
            p, _ = self.analyze_face(frame[y:y + h, x:x + w])
            if p is not None:                                       # backend_server.py:166
                self.tracker.update(p)
                return {'analysis_mode': 'face+frame', 'faces_detected': len(faces), 'fake_probability': p,
                        'face_probability': p, 'frame_forensic_probability': fprob, 'real_probability': 1 - p,
                        'confidence_level': self.tracker.confidence_level(), 'frame_count': self.frame_count,
                        'face_bbox': {'x': x, 'y': y, 'width': w, 'height': h}, 'votes': self.tracker.voting_stats()}
        self.tracker.update(fprob)
        return {'analysis_mode': 'frame_only', 'faces_detected': len(faces), 'fake_probability': fprob,
                'frame_forensic_probability': fprob, 'real_probability': 1 - fprob,
                'confidence_level': self.tracker.confidence_level(), 'frame_count': self.frame_count,
                'votes': self.tracker.voting_stats()}
