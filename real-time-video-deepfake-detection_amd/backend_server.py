"""Flask surface of the reference's ``backend_server.py``, unchanged for clients.

Routes, methods, status codes and JSON keys follow reference backend_server.py:82-255 (the wire
contract the Chrome extension reads): ``POST /analyze`` (multipart field
``frame``), ``GET /health``, ``POST /reset``, ``GET /stats``; 400 for a missing/undecodable
frame, 429 from the 100 ms rate limiter, 500 with ``{'error': ...}`` for anything else.  The
per-request work is `DeepfakeDetector.analyze_request` (one GPU call).  Differences: sequential-Huffman
JPEGs (what the extension sends) are decoded on the GPU from the request bytes (dfd_analyze_jpeg: entropy decoding
on the host, IDCT / upsampling / colour on the device, bit-identical to libjpeg) and everything else with
Pillow (cv2 is not a dependency here); ``POST /analyze_batch`` takes several ``frame`` parts of one stream in one
request; CORS headers are added by hand (no
flask_cors), and the rate limiter and detector are guarded by locks (the reference shares them
unsynchronised across Flask's worker threads, :57,62-80,275).
"""
from __future__ import annotations

import io
import logging
import threading
import time
import traceback
from functools import wraps

import numpy as np
from flask import Flask, jsonify, request

from . import runtime
from .deepfake_detection import DEVICE, DeepfakeDetector, model, mtcnn  # noqa: F401  (reference :39)
from .face_detection import detect_bounding_box  # noqa: F401                      (reference :40)

logging.basicConfig(level=logging.INFO, format='%(asctime)s [%(levelname)s] %(message)s', datefmt='%H:%M:%S')
logger = logging.getLogger(__name__)

app = Flask(__name__)
# cv2.imdecode has no request-size notion; Flask would buffer any body.  32 MiB holds a 4K PNG or a 32-frame JPEG batch.
app.config['MAX_CONTENT_LENGTH'] = 32 << 20
MAX_BATCH_FRAMES = 32                     # /analyze_batch: frames per request (one lock hold, one device batch)
# /analyze_batch: pixels per request, summed over the parts' HEADERS before anything is decoded or allocated (a flat
# 8192 x 8192 JPEG is ~1 MB: 32 of them fit the body limit and would ask for ~13 GB of pinned coefficients).  32 frames
# of 1080p; the library enforces its own budget too (dfd_common.h kMaxBatchPixels).
MAX_BATCH_PIXELS = 32 * 1920 * 1088
MAX_FRAME_PIXELS = 1 << 26                # one part (the GPU JPEG path's own cap; Pillow's bomb guard is of this order)


@app.after_request
def _cors(resp):                                  # reference :46-53 (CORS(app, origins="*", GET/POST/OPTIONS))
    resp.headers["Access-Control-Allow-Origin"] = "*"
    resp.headers["Access-Control-Allow-Methods"] = "GET, POST, OPTIONS"
    resp.headers["Access-Control-Allow-Headers"] = "Content-Type"
    return resp


detector = DeepfakeDetector(enable_gradcam=False, use_tta=False, num_tta_augmentations=1,
                            detection_threshold=0.55)                       # reference :57

_last_request_time = 0.0
_min_request_interval = 0.1                                                 # reference :63
_rate_lock = threading.Lock()
_detector_lock = threading.Lock()


def rate_limit(f):
    """reference :66-80"""
    @wraps(f)
    def decorated(*args, **kwargs):
        global _last_request_time
        with _rate_lock:
            now = time.time()
            elapsed = now - _last_request_time
            if elapsed < _min_request_interval:
                return jsonify({'error': 'Rate limited',
                                'retry_after_ms': int((_min_request_interval - elapsed) * 1000)}), 429
            _last_request_time = now
        return f(*args, **kwargs)
    return decorated


def image_size(image_bytes: bytes):
    """(width, height) from the file's header alone (nothing is decoded), or None."""
    from PIL import Image

    try:
        with Image.open(io.BytesIO(image_bytes)) as im:
            return im.size
    except Exception:
        return None


def decode_image(image_bytes: bytes):
    """cv2.imdecode(..., IMREAD_COLOR) stand-in: BGR uint8 (H,W,3) or None (reference :139-145)."""
    from PIL import Image

    try:
        with Image.open(io.BytesIO(image_bytes)) as im:
            if im.size[0] * im.size[1] > MAX_FRAME_PIXELS:
                return None
            rgb = np.asarray(im.convert("RGB"))
    except Exception:
        return None
    if rgb.ndim != 3 or rgb.shape[0] < 1 or rgb.shape[1] < 1:
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _gpu_name():
    try:
        import torch

        return torch.cuda.get_device_name(runtime.device_index()) if torch.cuda.is_available() else None
    except Exception:
        return None


@app.route('/health', methods=['GET'])
def health_check():
    """reference :82-99"""
    name = _gpu_name()
    h = runtime.peek_default_handle()
    # reference :93: face detection is the DNN or its Haar fallback (face_detection.py:58-61)
    has_det = bool(h.has_detector or h.has_haar) if h is not None else bool(
        runtime.detector_loaded or runtime.detector_synthetic or runtime.haar_loaded)
    # the reference's keys (:84-99) plus what was actually loaded: without trained weights the verdicts mean nothing
    return jsonify({'status': 'healthy', 'model_loaded': bool(runtime.model_loaded), 'detector_loaded': bool(runtime.detector_loaded),
                    'mtcnn_loaded': bool(runtime.mtcnn_loaded), 'synthetic_weights': bool(runtime.detector_synthetic),
                    'device': DEVICE, 'gpu_name': name, 'frame_count': detector.frame_count,
                    'capabilities': {'face_detection': has_det, 'frame_forensics': True, 'temporal_tracking': True}}), 200


@app.route('/reset', methods=['POST'])
def reset_detector():
    """reference :101-115"""
    try:
        with _detector_lock:
            detector.reset()
        return jsonify({'success': True, 'message': 'Detector reset successfully'}), 200
    except Exception as e:
        logger.error("Reset failed: %s", e)
        return jsonify({'success': False, 'error': str(e)}), 500


@app.route('/analyze', methods=['POST'])
@rate_limit
def analyze_frame():
    """reference :117-238"""
    start_time = time.time()
    try:
        if 'frame' not in request.files:
            return jsonify({'error': 'No frame provided'}), 400
        image_bytes = request.files['frame'].read()
        response = None
        if image_bytes[:2] == b'\xff\xd8':                        # JPEG: decode on the device, no raw upload
            try:
                with _detector_lock:
                    response = detector.analyze_request(jpeg=image_bytes)
            except runtime.DfdError as e:
                if e.code not in (-7, -1):                          # unsupported flavour / not decodable here: Pillow decides
                    raise
        if response is None:
            frame = decode_image(image_bytes)
            if frame is None:
                return jsonify({'error': 'Invalid image format'}), 400
            with _detector_lock:
                response = detector.analyze_request(frame)
        ms = (time.time() - start_time) * 1000
        # key order as the reference builds it (:178-195 / :213-225)
        bbox = response.pop('face_bbox', None)
        response['processing_time_ms'] = round(ms, 1)
        if bbox is not None:
            response['face_bbox'] = bbox
        logger.info("Frame %d%s | %.0f%% | Verdict: %s | %.0fms", response['frame_count'],
                    "" if bbox else " [NO FACE]", response['fake_probability'] * 100, response['confidence_level'], ms)
        return jsonify(response), 200
    except Exception as e:
        logger.error("Error analyzing frame: %s", e)
        logger.error(traceback.format_exc())
        return jsonify({'error': str(e)}), 500


@app.route('/analyze_batch', methods=['POST'])
@rate_limit
def analyze_batch():
    """Several consecutive frames of ONE stream in a single request (multipart parts all named ``frame``, in stream
    order): the same per-frame flow and vote order as /analyze, one JSON object per frame under ``results``.  Not in the
    reference (its extension posts one frame per request, throttled to 10 per second, backend_server.py:62-80); this
    is the batched entry SURVEY 8(f) N2 asks for.  Every part is read and validated BEFORE the detector is touched (a
    rejected request leaves the stream's state alone); then ONE library call does the GPU work of all frames
    (`DeepfakeDetector.analyze_request_batch`: JPEG scans entropy-decoded in parallel on host threads, one forensic /
    detector / classifier pass) and the votes are replayed in order.  At most MAX_BATCH_FRAMES parts."""
    start_time = time.time()
    try:
        files = request.files.getlist('frame')
        if not files:
            return jsonify({'error': 'No frame provided'}), 400
        if len(files) > MAX_BATCH_FRAMES:
            return jsonify({'error': f'Too many frames in one request ({len(files)} > {MAX_BATCH_FRAMES})'}), 400
        blobs = [f.read() for f in files]
        pixels = 0
        for k, data in enumerate(blobs):                            # headers only: the budget is checked before any decode
            size = image_size(data)
            if size is None:
                return jsonify({'error': f'Invalid image format (frame {k})'}), 400
            pixels += size[0] * size[1]
        if pixels > MAX_BATCH_PIXELS:
            return jsonify({'error': f'Batch too large ({pixels} pixels > {MAX_BATCH_PIXELS})'}), 400
        items = []
        for k, data in enumerate(blobs):
            if data[:2] == b'\xff\xd8':
                items.append(data)                                  # decoded on the device
            else:
                frame = decode_image(data)
                if frame is None:
                    return jsonify({'error': f'Invalid image format (frame {k})'}), 400
                items.append(frame)
        results = None
        with _detector_lock:
            try:
                results = detector.analyze_request_batch(items)
            except (runtime.DfdError, ValueError) as e:
                # a JPEG flavour the device path does not take, or parts of different sizes: nothing has moved yet
                if isinstance(e, runtime.DfdError) and e.code not in (-7, -1):
                    raise
        if results is None:
            frames = []
            for k, it in enumerate(items):
                frame = decode_image(it) if isinstance(it, bytes) else it
                if frame is None:
                    return jsonify({'error': f'Invalid image format (frame {k})'}), 400
                frames.append(frame)
            with _detector_lock:
                if len({fr.shape for fr in frames}) == 1:
                    results = detector.analyze_request_batch(frames)
                else:
                    results = [detector.analyze_request(fr) for fr in frames]
        ms = (time.time() - start_time) * 1000
        return jsonify({'success': True, 'frames': len(results), 'processing_time_ms': round(ms, 1), 'results': results}), 200
    except Exception as e:
        logger.error("Error analyzing batch: %s", e)
        logger.error(traceback.format_exc())
        return jsonify({'error': str(e)}), 500


@app.route('/stats', methods=['GET'])
def get_stats():
    """reference :240-255"""
    try:
        t = detector.temporal_tracker
        return jsonify({'frame_count': detector.frame_count, 'temporal_average': float(t.get_temporal_average()),
                        'stability_score': float(t.get_stability_score()), 'confidence_level': t.get_confidence_level(),
                        'history_length': len(t.score_history), 'voting': t.get_voting_stats(), 'device': DEVICE}), 200
    except Exception as e:
        return jsonify({'error': str(e)}), 500


if __name__ == '__main__':
    app.run(host='0.0.0.0', port=5000, debug=False, threaded=True)         # reference :275
