"""Process-wide default library handle (the counterpart of the reference's import-time
``model = DeepfakeEfficientNet(...); model.to(DEVICE).eval()`` singleton,
reference deepfake_detection.py:30-90), created lazily on first use.

Weights: ``$DFD_WEIGHTS`` or ``weights/best_model.pth`` next to the package if present
(reference checkpoint layout), else the seeded random-init state dict - the reference tree ships
no weights.  Device: ``$DFD_DEVICE`` or ``$LOCAL_RANK`` or 0.
"""
from __future__ import annotations

import logging
import os
import threading
from typing import Dict, Optional

import numpy as np

from . import weights as W
from ._lib import Handle

log = logging.getLogger(__name__)
_lock = threading.Lock()
_default: Optional[Handle] = None
_state: Optional[Dict[str, np.ndarray]] = None
model_loaded = False          # True when a trained checkpoint was found


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


def default_handle() -> Handle:
    global _default
    with _lock:
        if _default is None:
            # detector: the reference's Caffe files are not in its tree either (face_detection.py:19-20);
            # seeded random-init weights of the same topology stand in
            seed = int(os.environ.get("DFD_SEED", "0"))
            ssd = W.seeded_ssd_state_dict(seed)
            # MTCNN: facenet-pytorch ships its trained pnet/rnet/onet inside the pip package, which is absent here;
            # $DFD_MTCNN_WEIGHTS = directory with pnet.pt / rnet.pt / onet.pt, else seeded random-init; DFD_MTCNN=0
            # leaves the stage out (the detector crop then feeds the classifier directly)
            mt = None
            if os.environ.get("DFD_MTCNN", "1") != "0":
                d = os.environ.get("DFD_MTCNN_WEIGHTS")
                mt = W.load_mtcnn_checkpoints(d) if d else W.seeded_mtcnn_state_dict(seed)
            _default = Handle(W.pack_all(default_state_dict(), ssd, mt), device=device_index(),
                              max_batch=int(os.environ.get("DFD_MAX_BATCH", "16")))
        return _default


def peek_default_handle() -> Optional[Handle]:
    """The default handle if it has been created, without creating it."""
    return _default


def set_default_handle(h: Optional[Handle]) -> None:
    global _default
    with _lock:
        _default = h
