"""Host mirror of the reference's ``deepfake_detection.py`` over the HIP path.

`DeepfakeDetector` keeps the constructor, attributes, methods and result dictionaries of
reference deepfake_detection.py:292-726 and the module keeps its globals (`DEVICE`, `model`,
`mtcnn`, `detector`, `predict`, `predict_with_forensics`; :21-32,729-747).  Per frame the GPU
does forensics, face detection, crop -> CLAHE -> 224x224 -> EfficientNet-B0 in ONE call
(`dfd_analyze_frame`); calibration, the small-face heuristic and the vote stay on the host as in
the reference.  The MTCNN align/crop of reference :376-380 runs inside the same library call when the handle's
weights carry the cascade (`mtcnn` below is its module-level mirror); a crop in which it finds no face yields no
prediction, exactly as the reference's `None` (:378-380, :616-617, backend_server.py:166).  Deliberate differences
(DESIGN.md section 8): GradCAM is not implemented (disabled in the reference's shipped configurations, :730-736 and
backend_server.py:57), frames are returned un-annotated, nothing is printed per frame.  TTA (:408-443) is
`analyze_face_with_tta` over `dfd_tta_augment`.  Face detection inside the fused call follows the reference's
`detect_bounding_box` (face_detection.py:58-66): the SSD when the handle carries one, else - and after an SSD failure -
the Haar cascade, else no faces ('frame_only').
"""
from __future__ import annotations

import logging
import os
import pickle
import threading
from typing import List, Optional

import numpy as np

from . import runtime
from ._lib import DfdError, Handle
from .face_detection import detect_bounding_box
from .frame_analysis import FrameForensicAnalyzer
from .tracker import TemporalTracker

log = logging.getLogger(__name__)

DEVICE = f"cuda:{runtime.device_index()}"
from .mtcnn import MTCNN  # noqa: E402

mtcnn = MTCNN(select_largest=False, post_process=False, device=DEVICE)     # reference :24-28


def _sigmoid32(logit) -> float:
    """torch.sigmoid on a float32 scalar, then .item() (reference :397-398)."""
    x = np.float32(logit)
    return float(np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32)))


class _LazyModel:
    """`model` global of the reference (:32): the classifier bound to the default handle."""

    def __call__(self, rgb_input, freq_input=None):
        x = rgb_input.detach().cpu().numpy() if hasattr(rgb_input, "detach") else np.asarray(rgb_input)
        out = runtime.default_handle().classify(np.ascontiguousarray(x, dtype=np.float32))
        if hasattr(rgb_input, "detach"):
            import torch

            return torch.from_numpy(out)
        return out

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self


model = _LazyModel()


class DeepfakeDetector:
    def __init__(self, enable_gradcam=False, use_tta=True, num_tta_augmentations=3, detection_threshold=0.5,
                 face_weight=0.70, forensic_weight=0.30, *, handle: Optional[Handle] = None):
        self.enable_gradcam = enable_gradcam
        self.use_tta = use_tta
        self.num_tta_augmentations = num_tta_augmentations
        self.detection_threshold = detection_threshold
        self.face_weight = face_weight                  # stored, never used - as in the reference
        self.forensic_weight = forensic_weight
        self.temporal_tracker = TemporalTracker(window_size=60, high_confidence_threshold=0.6, voting_window=10,
                                                detection_threshold=detection_threshold)
        self.frame_count = 0
        self._handle = handle
        self.frame_analyzer = FrameForensicAnalyzer(analysis_size=(256, 256), handle=handle)
        self.full_forensic_interval = 3
        self.last_frame_forensic_result = None
        self.calibrator = None
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights", "calibrator.pkl")
        if os.path.exists(path):
            try:
                with open(path, "rb") as f:
                    self.calibrator = pickle.load(f)
            except Exception:                             # reference :341-342
                log.warning("could not load calibrator")
        self._lock = threading.Lock()                     # a handle is single-caller; Flask is threaded

    # ------------------------------------------------------------------ plumbing
    @property
    def handle(self) -> Handle:
        if self._handle is None:
            self._handle = runtime.default_handle()
            self.frame_analyzer._handle = self._handle
        return self._handle

    def reset(self):
        """reference :344-355"""
        self.temporal_tracker.reset()
        self.frame_count = 0
        self.frame_analyzer.reset()
        self.last_frame_forensic_result = None

    # ------------------------------------------------------------------ scalars after the network
    def apply_calibration(self, raw_prob):
        """reference :445-455"""
        if self.calibrator is None:
            return raw_prob
        try:
            return self.calibrator.predict_proba([[raw_prob]])[0][1]
        except Exception:
            return raw_prob

    def apply_heuristics(self, fake_prob, face_region):
        """reference :489-502: +0.10 for crops under 80 px, clipped to [0,1]."""
        h, w = face_region.shape[:2]
        return self._heuristics_hw(fake_prob, h, w)

    @staticmethod
    def _heuristics_hw(fake_prob, h, w):
        adjustment = 0.10 if (h < 80 or w < 80) else 0.0
        return np.clip(fake_prob + adjustment, 0, 1)

    def _finish_face(self, logit, h, w):
        """None when the MTCNN stage found no face in the crop (NaN logit from the library)."""
        if logit is None or np.isnan(logit):
            return None
        p = self.apply_calibration(_sigmoid32(logit))
        return self._heuristics_hw(p, h, w)

    # ------------------------------------------------------------------ reference methods
    def preprocess_face_quality(self, face_region):
        """BGR -> Lab, CLAHE(2.0, 8x8) on L, Lab -> BGR on the GPU (reference :357-370)."""
        with self._lock:
            return self.handle.preprocess_face_quality(np.ascontiguousarray(face_region))

    def _forensic_is_full(self) -> bool:
        return self.frame_count % self.full_forensic_interval == 0          # reference :509

    def analyze_frame_forensics(self, frame):
        """reference :504-515"""
        with self._lock:
            if self._forensic_is_full():
                result = self.frame_analyzer.analyze(frame)
            else:
                result = self.frame_analyzer.analyze_fast(frame)
        self.last_frame_forensic_result = result
        return result

    def _single_prediction(self, face_region):
        """sigmoid(model(...)) of an already quality-preprocessed crop, or None (reference :372-406)"""
        h, w = face_region.shape[:2]
        with self._lock:
            logit = self.handle.classify_crops(face_region, [(0, 0, w, h)], apply_clahe=False)[0, 0]
        return None if np.isnan(logit) else _sigmoid32(logit)

    def analyze_face_with_tta(self, face_region):
        """reference :408-443: the original plus num_tta_augmentations - 1 randomly augmented copies (horizontal flip
        with probability 1/2, brightness x U(0.9, 1.1), rotation by U(-3, 3) degrees about the centre), each through
        `_single_prediction`, averaged.  The draws come from Python's `random` in the reference's order."""
        import random

        predictions = []
        pred = self._single_prediction(face_region)
        if pred is not None:
            predictions.append(pred)
        for _ in range(self.num_tta_augmentations - 1):
            flip = random.random() > 0.5
            brightness = random.uniform(0.9, 1.1)
            angle = random.uniform(-3, 3)
            with self._lock:
                aug = self.handle.tta_augment(face_region, flip, brightness, angle)
            pred = self._single_prediction(aug)
            if pred is not None:
                predictions.append(pred)
        return float(np.mean(predictions)) if predictions else None

    def analyze_face(self, face_region):
        """(fake_prob, fake_prob, None) or (None, None, None) (reference :517-550)."""
        try:
            face = np.ascontiguousarray(face_region)
            if face.ndim != 3 or face.shape[2] != 3 or face.shape[0] < 1 or face.shape[1] < 1:
                return None, None, None
            h, w = face.shape[:2]
            if self.use_tta and self.num_tta_augmentations > 1:               # reference :524-527
                with self._lock:
                    pre = self.handle.preprocess_face_quality(face)
                raw = self.analyze_face_with_tta(pre)
                if raw is None:
                    return None, None, None
                p = self._heuristics_hw(self.apply_calibration(raw), h, w)
                return p, p, None
            with self._lock:
                logit = self.handle.classify_crops(face, [(0, 0, w, h)], apply_clahe=True)[0, 0]
            p = self._finish_face(logit, h, w)
            if p is None:
                return None, None, None
            return p, p, None
        except (DfdError, ValueError) as e:
            log.warning("face analysis error: %s", e)
            return None, None, None

    def _frame_on_gpu(self, frame, max_faces, jpeg: Optional[bytes] = None):
        """forensics + detection + per-face logits in one library call.  With `jpeg` the frame is decoded on the
        device from the request's bytes (dfd_analyze_jpeg) instead of uploaded raw; returns its (H, W) as 4th item."""
        full = self._forensic_is_full()
        with self._lock:
            if jpeg is not None:
                scores, prob, boxes, logits, shape = self.handle.analyze_jpeg(
                    jpeg, full, stream_id=self.frame_analyzer.stream_id, confidence_threshold=0.5, max_faces=max_faces)
                number = self.frame_analyzer.frame_count
                forensic = {'scores': scores, 'fake_probability': prob,
                            'analysis_type': 'frame_forensic' if full else 'frame_forensic_fast', 'frame_number': number}
                self.last_frame_forensic_result = forensic
                return forensic, boxes, logits, shape
            if self.handle.has_detector or self.handle.has_haar:
                # SSD, or the reference's Haar fallback (face_detection.py:58-61), inside the same library call
                scores, prob, boxes, logits = self.handle.analyze_frame(
                    frame, full, stream_id=self.frame_analyzer.stream_id, confidence_threshold=0.5, max_faces=max_faces)
            else:                                   # no detector of either kind: 'frame_only' mode (runtime.py)
                scores, prob, _ = self.handle.forensics(frame, full=full, stream_id=self.frame_analyzer.stream_id)
                boxes, logits = [], []
            number = self.frame_analyzer.frame_count
        forensic = {'scores': scores, 'fake_probability': prob,
                    'analysis_type': 'frame_forensic' if full else 'frame_forensic_fast', 'frame_number': number}
        self.last_frame_forensic_result = forensic
        return forensic, boxes, logits

    def predict(self, frame):
        """(frame, trigger_forensic, forensic_frame, result_data) (reference :588-686)."""
        self.frame_count += 1
        frame = np.ascontiguousarray(frame)
        small = frame.shape[0] < 30 or frame.shape[1] < 30
        frame_forensic, faces, logits = self._frame_on_gpu(frame, max_faces=200)       # DetectionOutput keeps <= 200
        if small:
            faces, logits = [], []
        trigger_forensic, forensic_frame = False, None
        face_results: List[dict] = []
        confidence_level = self.temporal_tracker.get_confidence_level()
        if len(faces) > 0:
            tta = self.use_tta and self.num_tta_augmentations > 1
            for (x, y, w, h), logit in zip(faces, logits):
                # with TTA every face goes through analyze_face (augmented copies), as the reference does (:614)
                fake_prob = self.analyze_face(frame[y:y + h, x:x + w])[0] if tta else self._finish_face(logit, h, w)
                if fake_prob is None:                                        # reference :616-617
                    continue
                self.temporal_tracker.update(fake_prob)
                confidence_level = self.temporal_tracker.get_confidence_level()
                if self.temporal_tracker.should_trigger_forensic_analysis():
                    trigger_forensic, forensic_frame = True, frame.copy()
                face_results.append({'face_prob': float(fake_prob), 'combined_prob': float(fake_prob),
                                     'bbox': {'x': int(x), 'y': int(y), 'w': int(w), 'h': int(h)}})
        else:
            self.temporal_tracker.update(frame_forensic['fake_probability'])
            confidence_level = self.temporal_tracker.get_confidence_level()
            if self.temporal_tracker.should_trigger_forensic_analysis():
                trigger_forensic, forensic_frame = True, frame.copy()
        result_data = {
            'frame_count': self.frame_count,
            'faces_detected': len(faces),
            'face_results': face_results,
            'frame_forensic': frame_forensic,
            'confidence_level': confidence_level if faces or self.frame_count > 1 else 'UNCERTAIN',
            'temporal_average': float(self.temporal_tracker.get_temporal_average()),
            'stability_score': float(self.temporal_tracker.get_stability_score()),
            'analysis_mode': 'face+frame' if len(faces) > 0 else 'frame_only',
        }
        return frame, trigger_forensic, forensic_frame, result_data

    def analyze(self, frame):
        """`result_data` of `predict` - the name BASELINE.json's north_star uses."""
        return self.predict(frame)[3]

    def analyze_request(self, frame=None, *, jpeg: Optional[bytes] = None):
        """The /analyze flow of the reference server (backend_server.py:147-233): forensics BEFORE the
        frame counter moves, only faces[0] is classified, one vote per request.  Returns the response
        dict without timing.  `jpeg`: the request's bytes, decoded on the device; raises
        DfdError (code -7 / -1) when the library does not decode that file - the caller then passes a frame."""
        if jpeg is not None:
            if not (self.handle.has_detector or self.handle.has_haar):
                frame = self.handle.decode_jpeg(jpeg)              # frame_only mode has no fused JPEG entry point
                jpeg = None
            else:
                frame_forensic, faces, logits, shape = self._frame_on_gpu(None, max_faces=1, jpeg=jpeg)
                small = shape[0] < 30 or shape[1] < 30
        if jpeg is None:
            frame = np.ascontiguousarray(frame)
            small = frame.shape[0] < 30 or frame.shape[1] < 30
            frame_forensic, faces, logits = self._frame_on_gpu(frame, max_faces=1)
        n_detected = 0 if small else self._last_face_count(frame, faces)
        self.frame_count += 1
        tr = self.temporal_tracker
        fprob = frame_forensic['fake_probability']
        fake_prob = None
        if len(faces) > 0 and not small:
            x, y, w, h = faces[0]
            fake_prob = self._finish_face(logits[0], h, w)
        if fake_prob is not None:                                            # backend_server.py:166
            tr.update(fake_prob)
            return {'success': True, 'analysis_mode': 'face+frame', 'faces_detected': n_detected,
                    'fake_probability': float(fake_prob), 'face_probability': float(fake_prob),
                    'frame_forensic_probability': float(fprob), 'real_probability': float(1 - fake_prob),
                    'confidence_level': tr.get_confidence_level(), 'temporal_average': float(tr.get_temporal_average()),
                    'stability_score': float(tr.get_stability_score()), 'frame_count': self.frame_count,
                    'face_bbox': {'x': int(x), 'y': int(y), 'width': int(w), 'height': int(h)}}
        tr.update(fprob)
        return {'success': True, 'analysis_mode': 'frame_only', 'faces_detected': n_detected,
                'fake_probability': float(fprob), 'frame_forensic_probability': float(fprob),
                'real_probability': float(1 - fprob), 'confidence_level': tr.get_confidence_level(),
                'temporal_average': float(tr.get_temporal_average()), 'stability_score': float(tr.get_stability_score()),
                'frame_count': self.frame_count}

    def analyze_request_batch(self, items):
        """`analyze_request` for several consecutive frames of this stream in ONE library call (POST /analyze_batch):
        `items` are JPEG bytes and / or BGR frames of one size.  The GPU work of all frames is batched
        (dfd_analyze_stream_batch); the frame counter, the full / fast forensic schedule (reference
        deepfake_detection.py:509-512) and the votes (backend_server.py:166-176,205-211) advance frame by frame in
        request order, so the responses equal those of len(items) single requests.  Raises DfdError (-7 / -1) BEFORE
        any state has moved when a JPEG part is not decodable on the device path or the sizes differ."""
        n = len(items)
        full = [(self.frame_count + i) % self.full_forensic_interval == 0 for i in range(n)]
        with self._lock:
            if self.handle.has_detector or self.handle.has_haar:
                res, shape = self.handle.analyze_stream_batch(items, full, stream_id=self.frame_analyzer.stream_id,
                                                              confidence_threshold=0.5, max_faces=1)
            else:                                                   # no detector of either kind: forensics only, frame by frame
                res, shape = [], None
                for it, fl in zip(items, full):
                    fr = self.handle.decode_jpeg(it) if isinstance(it, (bytes, bytearray)) else np.ascontiguousarray(it)
                    scores, prob, _ = self.handle.forensics(fr, full=fl, stream_id=self.frame_analyzer.stream_id)
                    res.append((scores, prob, [], [], 0))
                    shape = fr.shape[:2]
            first_number = self.frame_analyzer.frame_count - n + 1
        out = []
        small = shape[0] < 30 or shape[1] < 30
        tr = self.temporal_tracker
        for i, (scores, fprob, faces, logits, n_detected) in enumerate(res):
            self.last_frame_forensic_result = {'scores': scores, 'fake_probability': fprob,
                                               'analysis_type': 'frame_forensic' if full[i] else 'frame_forensic_fast',
                                               'frame_number': first_number + i}
            self.frame_count += 1
            fake_prob = None
            if len(faces) > 0 and not small:
                x, y, w, h = faces[0]
                fake_prob = self._finish_face(logits[0], h, w)
            if fake_prob is not None:
                tr.update(fake_prob)
                out.append({'success': True, 'analysis_mode': 'face+frame', 'faces_detected': 0 if small else n_detected,
                            'fake_probability': float(fake_prob), 'face_probability': float(fake_prob),
                            'frame_forensic_probability': float(fprob), 'real_probability': float(1 - fake_prob),
                            'confidence_level': tr.get_confidence_level(), 'temporal_average': float(tr.get_temporal_average()),
                            'stability_score': float(tr.get_stability_score()), 'frame_count': self.frame_count,
                            'face_bbox': {'x': int(x), 'y': int(y), 'width': int(w), 'height': int(h)}})
                continue
            tr.update(fprob)
            out.append({'success': True, 'analysis_mode': 'frame_only', 'faces_detected': 0 if small else n_detected,
                        'fake_probability': float(fprob), 'frame_forensic_probability': float(fprob),
                        'real_probability': float(1 - fprob), 'confidence_level': tr.get_confidence_level(),
                        'temporal_average': float(tr.get_temporal_average()), 'stability_score': float(tr.get_stability_score()),
                        'frame_count': self.frame_count})
        return out

    def _last_face_count(self, frame, faces):
        """`faces_detected` of the server response counts ALL detections (backend_server.py:181); the fused call
        classified only the first and the library remembers how many there were."""
        if not faces:
            return 0
        return self.handle.last_detection_count()


_detector: Optional[DeepfakeDetector] = None
_detector_lock = threading.Lock()


def _global_detector() -> DeepfakeDetector:
    """reference :730-736: module-level instance, built on first use instead of at import."""
    global _detector
    with _detector_lock:
        if _detector is None:
            _detector = DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.5,
                                         face_weight=0.70, forensic_weight=0.30)
        return _detector


def __getattr__(name):
    if name == "detector":
        return _global_detector()
    raise AttributeError(name)


def predict(frame):
    """reference :739-742"""
    return _global_detector().predict(frame)[0]


def predict_with_forensics(frame):
    """reference :745-747"""
    return _global_detector().predict(frame)
