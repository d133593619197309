"""Host mirror of ``facenet_pytorch.MTCNN`` as the reference uses it (reference deepfake_detection.py:24-28:
``MTCNN(select_largest=False, post_process=False, device=DEVICE).to(DEVICE).eval()``, called as
``mtcnn(PIL_image)`` at :377).  The cascade runs in libdfd_hip.so (`dfd_mtcnn_align`, include/dfd_hip.h); only the
configuration the reference constructs is built - the package's defaults image_size 160, margin 0,
min_face_size 20, thresholds (0.6, 0.7, 0.7), factor 0.709, keep_all False - and other values are rejected.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from ._lib import Handle


class MTCNN:
    def __init__(self, image_size: int = 160, margin: int = 0, min_face_size: int = 20, thresholds=(0.6, 0.7, 0.7),
                 factor: float = 0.709, post_process: bool = True, select_largest: bool = True,
                 selection_method=None, keep_all: bool = False, device=None, *, handle: Optional[Handle] = None):
        if (image_size, margin, min_face_size, tuple(thresholds), factor) != (160, 0, 20, (0.6, 0.7, 0.7), 0.709):
            raise ValueError("only facenet-pytorch's default geometry (160 / 0 / 20 / .6,.7,.7 / .709) is built")
        if select_largest or keep_all or post_process or selection_method not in (None, "probability"):
            raise ValueError("only select_largest=False, keep_all=False, post_process=False (the reference's call) is built")
        self.image_size, self.margin, self.min_face_size = image_size, margin, min_face_size
        self.thresholds, self.factor = list(thresholds), factor
        self.post_process, self.select_largest, self.keep_all = post_process, select_largest, keep_all
        self.selection_method = "probability"
        self.device = device
        self._handle = handle

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    @property
    def handle(self) -> Handle:
        if self._handle is None:
            from . import runtime

            self._handle = runtime.default_handle()
        return self._handle

    @staticmethod
    def _as_bgr(img) -> np.ndarray:
        a = np.asarray(img)                      # PIL RGB image or (H, W, 3) uint8 RGB array, as the package accepts
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
            raise ValueError("MTCNN expects an RGB uint8 image (PIL or HxWx3 array)")
        return np.ascontiguousarray(a[..., ::-1])

    def forward(self, img, save_path=None, return_prob: bool = False):
        """(3, 160, 160) float RGB tensor in 0..255 (torch when importable, else numpy), or None."""
        if save_path is not None:
            raise ValueError("save_path is not supported")
        face, box = self.handle.mtcnn_align(self._as_bgr(img))
        if face is not None:
            try:
                import torch

                face = torch.from_numpy(face)
            except ImportError:
                pass
        if return_prob:
            return face, (None if box is None else float(box[4]))
        return face

    __call__ = forward

    def detect(self, img, landmarks: bool = False):
        """(boxes (n,4) float32, probs (n,)) of every face that passes the cascade, or (None, [None])."""
        if landmarks:
            raise ValueError("landmarks are not computed (they do not influence the crop)")
        rows = self.handle.mtcnn_tap(self._as_bgr(img), "stage3")
        if rows.shape[0] == 0:
            return None, [None]
        return rows[:, :4].copy(), rows[:, 4].copy()
