"""Process-wide default library handle (the counterpart of the reference's import-time
``model = DeepfakeEfficientNet(...); model.to(DEVICE).eval()`` singleton,
reference deepfake_detection.py:30-90), created lazily on first use.

Weights (the reference tree ships none):
  classifier  ``$DFD_WEIGHTS`` or ``weights/best_model.pth`` next to the package (reference checkpoint layout),
              else seeded random-init weights with a warning (`model_loaded` stays False, /health says so);
  detector    ``$DFD_SSD_WEIGHTS``: the reference's Caffe pair (face_detection.py:19-24) - a .caffemodel with its
              deploy.prototxt (``$DFD_SSD_PROTOTXT``, default: ``deploy.prototxt`` next to it), read by `caffe_io`
              into a detector plan - or an .npz / .pth state dict in `ssd_arch` naming.  Without it there is NO detector:
              the handle is built without one, `detect_bounding_box` returns [] and every frame is analysed in
              'frame_only' mode unless a Haar cascade is configured (the reference, whose model files are missing too,
              face_detection.py:22-34, falls back to Haar: next entry);
  Haar        ``$DFD_HAAR_CASCADE``: OpenCV's haarcascade_frontalface_default.xml (reference face_detection.py:12) - the
              detector the reference falls back to; used here when there are no SSD weights or the SSD call fails;
  MTCNN       ``$DFD_MTCNN_WEIGHTS``: directory with pnet.pt / rnet.pt / onet.pt (facenet-pytorch's files); without it
              the align stage is left out.
``DFD_SYNTHETIC_WEIGHTS=1`` (benchmarks, demos) substitutes seeded random-init detector and MTCNN weights of the same
topologies - boxes and alignments are then meaningless, and a warning says so.
Device: ``$DFD_DEVICE`` or ``$LOCAL_RANK`` or 0.
"""
from __future__ import annotations

import gc
import logging
import os
import threading
from typing import Dict, Optional

import numpy as np

from . import weights as W
from ._lib import DfdError, Handle  # noqa: F401  (re-exported: backend_server catches runtime.DfdError)

log = logging.getLogger(__name__)
_lock = threading.Lock()
_default: Optional[Handle] = None
_state: Optional[Dict[str, np.ndarray]] = None
model_loaded = False          # True when a trained classifier checkpoint was found
detector_loaded = False       # True when real detector weights were loaded ($DFD_SSD_WEIGHTS)
mtcnn_loaded = False          # True when real MTCNN weights were loaded ($DFD_MTCNN_WEIGHTS)
detector_synthetic = False    # seeded stand-ins in use (DFD_SYNTHETIC_WEIGHTS=1)
haar_loaded = False           # True when a Haar cascade XML was loaded ($DFD_HAAR_CASCADE)


def device_index() -> int:
    return int(os.environ.get("DFD_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def default_state_dict() -> Dict[str, np.ndarray]:
    global _state, model_loaded
    if _state is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.environ.get("DFD_WEIGHTS", os.path.join(here, "weights", "best_model.pth"))
        if os.path.exists(path):
            _state = W.load_checkpoint(path)
            model_loaded = True
            log.info("loaded checkpoint %s", path)
        else:
            _state = W.seeded_state_dict(int(os.environ.get("DFD_SEED", "0")))
            log.warning("no trained checkpoint (%s); using seeded random-init weights", path)
    return _state


def load_detector_weights(path: str):
    """-> (state dict, arch or None).  .caffemodel: (deploy.prototxt + caffemodel) through caffe_io -> its own layer
    plan; .npz / .pth: a state dict in ssd_arch naming (built-in topology)."""
    if path.endswith(".caffemodel"):
        from . import caffe_io

        proto = os.environ.get("DFD_SSD_PROTOTXT", os.path.join(os.path.dirname(path), "deploy.prototxt"))
        arch, sd = caffe_io.load_caffe_detector(proto, path)
        return sd, arch
    return _load_state_dict(path), None


def _load_state_dict(path: str) -> Dict[str, np.ndarray]:
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: np.asarray(z[k], np.float32) for k in z.files}
    import torch

    sd = torch.load(path, map_location="cpu")
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items()}


def default_handle() -> Handle:
    global _default, detector_loaded, mtcnn_loaded, detector_synthetic, haar_loaded
    with _lock:
        if _default is None:
            seed = int(os.environ.get("DFD_SEED", "0"))
            synthetic = os.environ.get("DFD_SYNTHETIC_WEIGHTS", "0") == "1"
            ssd, ssd_arch = None, None
            path = os.environ.get("DFD_SSD_WEIGHTS")
            if path:
                ssd, ssd_arch = load_detector_weights(path)
                detector_loaded = True
                log.info("loaded detector weights %s", path)
            elif synthetic:
                ssd = W.seeded_ssd_state_dict(seed)
                detector_synthetic = True
                log.warning("DFD_SYNTHETIC_WEIGHTS=1: random-init detector weights - face boxes are meaningless")
            else:
                log.warning("no detector weights ($DFD_SSD_WEIGHTS): faces come from the Haar cascade if $DFD_HAAR_CASCADE "
                            "is set, else every frame is analysed in 'frame_only' mode")
            mt = None
            if os.environ.get("DFD_MTCNN", "1") != "0":
                d = os.environ.get("DFD_MTCNN_WEIGHTS")
                if d:
                    mt = W.load_mtcnn_checkpoints(d)
                    mtcnn_loaded = True
                elif synthetic:
                    mt = W.seeded_mtcnn_state_dict(seed)
                    log.warning("DFD_SYNTHETIC_WEIGHTS=1: random-init MTCNN cascade - alignments are meaningless")
                else:
                    log.warning("no MTCNN weights ($DFD_MTCNN_WEIGHTS): the align stage is left out")
            hc = None
            cascade = os.environ.get("DFD_HAAR_CASCADE")
            if cascade:
                from . import haar

                hc = haar.load_cascade_xml(cascade)
                haar_loaded = True
                log.info("loaded Haar cascade %s (fallback detector)", cascade)
            _default = Handle(W.pack_all(default_state_dict(), ssd, mt, ssd_arch=ssd_arch, haar=hc), device=device_index(),
                              max_batch=int(os.environ.get("DFD_MAX_BATCH", "16")))
            # the start-up objects (state dicts, the blob, module globals) move to the permanent generation: a full
            # collection in the middle of a request stream otherwise walks them all - measured as a one-off 40 ms stall
            # of a 6.5 ms `analyze_batch` call (profiles/mtcnn_profile_driver.py, gone with the collector off)
            gc.collect()
            gc.freeze()
        return _default


def peek_default_handle() -> Optional[Handle]:
    """The default handle if it has been created, without creating it."""
    return _default


def set_default_handle(h: Optional[Handle]) -> None:
    global _default
    with _lock:
        _default = h
