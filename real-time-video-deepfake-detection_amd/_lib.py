"""ctypes binding of libdfd_hip.so (include/dfd_hip.h).

There is deliberately no fallback: if the shared library is missing, or no gfx950
device is visible, every compute entry point raises `DfdError`.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFD_LIB_PATH") or os.path.join(_HERE, "libdfd_hip.so")   # DFD_LIB_PATH: diagnostic builds only

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class DfdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libdfd_hip: {message} (status {code})")
        self.code = code


# name -> (restype, argtypes); the single source of truth the symbol-export test walks
SIGNATURES = {
    "dfd_abi_version": (C.c_int, []),
    "dfd_create": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "dfd_destroy": (None, [C.c_void_p]),
    "dfd_last_error": (C.c_char_p, [C.c_void_p]),
    "dfd_max_batch": (C.c_int, [C.c_void_p]),
    "dfd_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "dfd_gemm_tile_count": (C.c_int, []),
    "dfd_warmup": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "dfd_tiles_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dfd_tiles_import": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "dfd_gemm_chunk_rows": (C.c_longlong, [C.c_longlong, C.c_longlong, C.c_longlong]),
    "dfd_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dfd_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfd_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dfd_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dfd_sync": (C.c_int, [C.c_void_p]),
    "dfd_wait_for": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfd_frame_ptr": (C.c_void_p, [C.c_void_p]),
    "dfd_timer_begin": (C.c_int, [C.c_void_p]),
    "dfd_timer_end": (C.c_int, [C.c_void_p, c_float_p]),
    "dfd_classify_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dfd_classify_nchw_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dfd_extract_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dfd_b0_tap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p, C.c_size_t,
                              C.POINTER(C.c_size_t)]),
    "dfd_resize_bgr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dfd_tta_augment": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_void_p]),
    "dfd_preprocess_face_quality": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dfd_preprocess_crops": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p]),
    "dfd_classify_crops": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_void_p]),
    "dfd_frequency_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dfd_detect_faces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dfd_has_detector": (C.c_int, [C.c_void_p]),
    "dfd_last_detection_count": (C.c_int, [C.c_void_p]),
    "dfd_classifier_crop_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "dfd_ssd_tap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p,
                              C.c_size_t, C.POINTER(C.c_size_t)]),
    "dfd_has_haar": (C.c_int, [C.c_void_p]),
    "dfd_detect_faces_haar": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfd_has_mtcnn": (C.c_int, [C.c_void_p]),
    "dfd_mtcnn_align": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.POINTER(C.c_int)]),
    "dfd_mtcnn_tap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p,
                                C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p]),
    "dfd_analyze_frame": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "dfd_jpeg_coefficients": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_size_t)]),
    "dfd_decode_jpeg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    "dfd_decode_jpeg_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]),
    "dfd_jpeg_decode_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "dfd_analyze_jpeg": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]),
    "dfd_analyze_stream_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                           C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfd_analyze_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "dfd_forensics": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "dfd_forensics_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "dfd_forensics_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "dfd_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dfd_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dfd_analyze_frames_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "dfd_analyze_jpegs_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfd_forensic_signals_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "dfd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dfd_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "dfd_comm_destroy": (C.c_int, [C.c_void_p]),
    "dfd_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dfd_vote_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dfd_vote_allgather_waves": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "dfd_b0_profile_begin": (C.c_int, [C.c_void_p]),
    "dfd_b0_profile_end": (C.c_int, [C.c_void_p, c_float_p, C.POINTER(C.c_char_p), C.c_int,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load the shared library and declare every prototype.  Raises if it is not built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DfdError(-100, f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "or `make -C real-time-video-deepfake-detection_amd/csrc`")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def jpeg_coefficients(data: bytes):
    """Host half of the JPEG decoder (no GPU): -> dict(width, height, components, hmax, vmax, comps=[(bw, bh, tq)],
    qtables (4,64) uint16, coef int16 flat).  Raises DfdError (code -7 = flavour not decoded on the device path)."""
    lib = load()
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    info = (C.c_int * 16)()
    cnt = C.c_size_t()
    rc = lib.dfd_jpeg_coefficients(buf, len(data), info, None, None, 0, C.byref(cnt))
    if rc != 0:
        raise DfdError(rc, (lib.dfd_last_error(None) or b"").decode())
    q = np.zeros((4, 64), np.uint16)
    coef = np.zeros(cnt.value, np.int16)
    rc = lib.dfd_jpeg_coefficients(buf, len(data), info, _ptr(q), _ptr(coef), coef.size, C.byref(cnt))
    if rc != 0:
        raise DfdError(rc, (lib.dfd_last_error(None) or b"").decode())
    n = info[2]
    return {"width": info[0], "height": info[1], "components": n, "hmax": info[3], "vmax": info[4],
            "comps": [(info[5 + 3 * c], info[6 + 3 * c], info[7 + 3 * c]) for c in range(n)], "qtables": q, "coef": coef}


class DeviceBuffer:
    """A raw HBM allocation owned by a Handle."""

    def __init__(self, handle: "Handle", nbytes: int):
        self._h = handle
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        handle._check(handle._lib.dfd_device_alloc(handle._p, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        if a.nbytes > self.nbytes:
            raise ValueError("upload larger than buffer")
        self._h._check(self._h._lib.dfd_memcpy_h2d(self._h._p, self.ptr, _ptr(a), a.nbytes))
        return self

    def download(self, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than buffer")
        self._h._check(self._h._lib.dfd_memcpy_d2h(self._h._p, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self._h._lib.dfd_device_free(self._h._p, self.ptr)
            self.ptr = None


class Handle:
    """One (device, stream) context of the library: weights + workspaces for `max_batch` crops."""

    def __init__(self, blob: bytes, device: int = 0, max_batch: int = 8):
        self._lib = load()
        self._blob = blob                      # keep alive during create
        p = C.c_void_p()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        rc = self._lib.dfd_create(int(device), buf, len(blob), int(max_batch), C.byref(p))
        if rc != 0:
            msg = self._lib.dfd_last_error(None)
            raise DfdError(rc, (msg or b"dfd_create failed").decode())
        self._p = p
        self.device = int(device)
        self.max_batch = int(max_batch)

    # -- plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise DfdError(rc, (self._lib.dfd_last_error(self._p) or b"").decode())

    def close(self):
        if getattr(self, "_p", None):
            self._lib.dfd_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        self._check(self._lib.dfd_set_option(self._p, name.encode(), int(value)))

    def warmup(self, n_crops: int = 0, n_frames: int = 0):
        """Measure the split-GEMM tiles for these batch sizes (synchronises; serving calls never do).
        With DFD_TILE_CACHE=<file> in the environment the measured tiles are loaded from / saved to that file, so
        that e.g. a run under rocprofv3 launches no tuning candidates."""
        cache = os.environ.get("DFD_TILE_CACHE")
        if cache and os.path.exists(cache):
            self.tiles_import(open(cache).read())
        self._check(self._lib.dfd_warmup(self._p, int(n_crops), int(n_frames)))
        if cache:
            try:
                with open(cache, "w") as f:
                    f.write(self.tiles_export())
            except OSError:
                pass

    def tiles_export(self) -> str:
        n = C.c_size_t()
        self._check(self._lib.dfd_tiles_export(self._p, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        self._check(self._lib.dfd_tiles_export(self._p, buf, n.value, C.byref(n)))
        return buf.raw[: n.value].decode()

    def tiles_import(self, text: str) -> int:
        acc = C.c_int()
        b = text.encode()
        self._check(self._lib.dfd_tiles_import(self._p, b, len(b), C.byref(acc)))
        return acc.value

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def sync(self):
        self._check(self._lib.dfd_sync(self._p))

    def wait_for(self, other: "Handle"):
        """work queued on this handle from now on starts after what `other` has queued so far (no host wait)"""
        self._check(self._lib.dfd_wait_for(self._p, other._p))

    def timer_begin(self):
        self._check(self._lib.dfd_timer_begin(self._p))

    def timer_end(self) -> float:
        ms = C.c_float()
        self._check(self._lib.dfd_timer_end(self._p, C.byref(ms)))
        return float(ms.value)

    # -- classifier
    @staticmethod
    def _as_nchw(x) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        if a.ndim != 4 or a.shape[1:] != (3, 224, 224):
            raise ValueError(f"expected (B,3,224,224) float input, got {a.shape}")
        return a

    def classify(self, x) -> np.ndarray:
        a = self._as_nchw(x)
        out = np.empty((a.shape[0], 1), dtype=np.float32)
        self._check(self._lib.dfd_classify_nchw(self._p, _ptr(a), a.shape[0], _ptr(out)))
        return out

    def classify_device(self, x_dev: int, n: int, logits_dev: int):
        self._check(self._lib.dfd_classify_nchw_device(self._p, x_dev, int(n), logits_dev))

    def extract_features(self, x) -> np.ndarray:
        a = self._as_nchw(x)
        out = np.empty((a.shape[0], 1280), dtype=np.float32)
        self._check(self._lib.dfd_extract_features(self._p, _ptr(a), a.shape[0], _ptr(out)))
        return out

    def tap(self, x_dev: int, n: int, name: str, capacity: int) -> np.ndarray:
        out = np.empty(int(capacity), dtype=np.float32)
        cnt = C.c_size_t()
        self._check(self._lib.dfd_b0_tap(self._p, x_dev, int(n), name.encode(), _ptr(out), out.size, C.byref(cnt)))
        return out[: cnt.value]

    def profile_begin(self):
        self._check(self._lib.dfd_b0_profile_begin(self._p))

    def profile_end(self, max_layers: int = 128):
        """-> (steps, [(launch name, summed ms over those steps), ...])"""
        ms = (C.c_float * max_layers)()
        names = (C.c_char_p * max_layers)()
        cnt, steps = C.c_int(), C.c_int()
        self._check(self._lib.dfd_b0_profile_end(self._p, ms, names, max_layers, C.byref(cnt), C.byref(steps)))
        return steps.value, [(names[i].decode(), float(ms[i])) for i in range(cnt.value)]

    # -- 8-bit image path
    @staticmethod
    def _as_bgr(frame) -> np.ndarray:
        a = np.asarray(frame)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"expected (H,W,3) uint8 BGR image, got {a.dtype} {a.shape}")
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        return a

    @staticmethod
    def _as_boxes(boxes) -> np.ndarray:
        b = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))
        if b.shape[0] == 0:
            raise ValueError("no boxes")
        return b

    def resize_bgr(self, frame, dw: int, dh: int) -> np.ndarray:
        a = self._as_bgr(frame)
        out = np.empty((dh, dw, 3), np.uint8)
        self._check(self._lib.dfd_resize_bgr(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], dh, dw, _ptr(out)))
        return out

    def preprocess_face_quality(self, face) -> np.ndarray:
        a = self._as_bgr(face)
        out = np.empty_like(a)
        self._check(self._lib.dfd_preprocess_face_quality(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], _ptr(out)))
        return out

    def tta_augment(self, face, flip: bool, brightness: float, angle_deg: float) -> np.ndarray:
        """cv2.flip / convertScaleAbs / warpAffine chain of the reference's test-time augmentation, on the device"""
        a = self._as_bgr(face)
        out = np.empty_like(a)
        self._check(self._lib.dfd_tta_augment(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], int(bool(flip)),
                                              float(brightness), float(angle_deg), _ptr(out)))
        return out

    def preprocess_crops(self, frame, boxes, apply_clahe: bool = True) -> np.ndarray:
        a, b = self._as_bgr(frame), self._as_boxes(boxes)
        out = np.empty((b.shape[0], 3, 224, 224), np.float32)
        self._check(self._lib.dfd_preprocess_crops(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], _ptr(b),
                                                   b.shape[0], int(apply_clahe), _ptr(out)))
        return out

    def classify_crops(self, frame, boxes, apply_clahe: bool = True) -> np.ndarray:
        a, b = self._as_bgr(frame), self._as_boxes(boxes)
        out = np.empty((b.shape[0], 1), np.float32)
        self._check(self._lib.dfd_classify_crops(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], _ptr(b),
                                                 b.shape[0], int(apply_clahe), _ptr(out)))
        return out

    # -- frame forensics
    FORENSIC_KEYS = ("frequency", "noise", "ela", "edge", "color", "temporal")
    FORENSIC_STAT_KEYS = ("freq_low", "freq_mid", "freq_high", "freq_high_ratio", "freq_mid_ratio", "freq_mid_cv",
                          "noise_mean", "noise_cv", "ela_mean", "ela_cv", "edge_density", "lap_var", "sat_std",
                          "val_std", "unique_hues", "mean_diff", "temporal_cv", "frame_count")

    def forensics(self, frame, full: bool = True, stream_id: int = 0):
        """-> (scores dict, fake_probability, stats dict)"""
        a = self._as_bgr(frame)
        sc = np.empty(6, np.float64)
        st = np.empty(len(self.FORENSIC_STAT_KEYS), np.float64)
        prob = C.c_double()
        self._check(self._lib.dfd_forensics(self._p, int(stream_id), _ptr(a), a.shape[0], a.shape[1], a.strides[0],
                                            int(bool(full)), _ptr(sc), C.byref(prob), _ptr(st)))
        scores = {k: float(v) for k, v in zip(self.FORENSIC_KEYS, sc) if not np.isnan(v)}
        return scores, float(prob.value), dict(zip(self.FORENSIC_STAT_KEYS, (float(v) for v in st)))

    def forensics_reset(self, stream_id: int = 0):
        self._check(self._lib.dfd_forensics_reset(self._p, int(stream_id)))

    def forensics_state(self, stream_id: int = 0):
        fc, nd, hp = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.dfd_forensics_state(self._p, int(stream_id), C.byref(fc), C.byref(nd), C.byref(hp)))
        return fc.value, nd.value, bool(hp.value)

    def frequency_features(self, image) -> np.ndarray:
        a = np.ascontiguousarray(np.asarray(image))
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
            raise ValueError(f"expected (H,W,3) or (H,W) uint8 image, got {a.dtype} {a.shape}")
        out = np.empty((2, 224, 224), np.float32)
        self._check(self._lib.dfd_frequency_features(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0],
                                                     3 if a.ndim == 3 else 1, _ptr(out)))
        return out

    # -- face detector
    @property
    def has_detector(self) -> bool:
        return bool(self._lib.dfd_has_detector(self._p))

    def detect_faces(self, frame, confidence_threshold: float = 0.5, max_out: int = 200, with_conf: bool = False):
        """-> [(x, y, w, h), ...] in descending-confidence order (reference face_detection.py:71-105)."""
        a = self._as_bgr(frame)
        boxes = np.zeros((max_out, 4), np.int32)
        conf = np.zeros(max_out, np.float32)
        n = C.c_int()
        self._check(self._lib.dfd_detect_faces(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0],
                                               float(confidence_threshold), _ptr(boxes), _ptr(conf), max_out, C.byref(n)))
        out = [tuple(int(v) for v in boxes[i]) for i in range(n.value)]
        return (out, conf[: n.value].copy()) if with_conf else out

    @property
    def has_haar(self) -> bool:
        return bool(self._lib.dfd_has_haar(self._p))

    def detect_faces_haar(self, frame, scale_factor: float = 1.1, min_neighbors: int = 5, min_size: int = 30,
                          max_out: int = 256, with_candidates: bool = False):
        """cv2 detectMultiScale on the blob's Haar cascade -> [(x, y, w, h), ...] (reference face_detection.py:108-123)"""
        a = self._as_bgr(frame)
        boxes = np.zeros((max_out, 4), np.int32)
        n, nc = C.c_int(), C.c_int()
        self._check(self._lib.dfd_detect_faces_haar(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], float(scale_factor),
                                                    int(min_neighbors), int(min_size), _ptr(boxes), max_out, C.byref(n), C.byref(nc)))
        out = [tuple(int(v) for v in boxes[i]) for i in range(n.value)]
        return (out, nc.value) if with_candidates else out

    def analyze_frame(self, frame, full_forensics: bool, stream_id: int = 0, confidence_threshold: float = 0.5,
                      max_faces: int = 16, apply_clahe: bool = True):
        """One upload: forensics + detect + crop/CLAHE + classify.
        -> (scores dict, forensic probability, [(x,y,w,h)...], logits (n,))"""
        a = self._as_bgr(frame)
        max_faces = max(1, int(max_faces))
        sc = np.empty(6, np.float64)
        prob = C.c_double()
        boxes = np.zeros((max_faces, 4), np.int32)
        logits = np.zeros(max_faces, np.float32)
        n = C.c_int()
        self._check(self._lib.dfd_analyze_frame(self._p, int(stream_id), _ptr(a), a.shape[0], a.shape[1], a.strides[0],
                                                int(bool(full_forensics)), float(confidence_threshold), max_faces,
                                                int(bool(apply_clahe)), _ptr(sc), C.byref(prob), _ptr(boxes), C.byref(n),
                                                _ptr(logits)))
        scores = {k: float(v) for k, v in zip(self.FORENSIC_KEYS, sc) if not np.isnan(v)}
        return scores, float(prob.value), [tuple(int(v) for v in boxes[i]) for i in range(n.value)], logits[: n.value].copy()

    UNSUPPORTED = -7

    def decode_jpeg(self, data: bytes) -> np.ndarray:
        """cv2.imdecode(IMREAD_COLOR) for a sequential-Huffman JPEG -> (H,W,3) uint8 BGR.  DfdError with
        .code == Handle.UNSUPPORTED for flavours the device path does not decode."""
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        hh, ww = C.c_int(), C.c_int()
        self._check(self._lib.dfd_decode_jpeg(self._p, buf, len(data), None, 0, C.byref(hh), C.byref(ww)))
        out = np.empty((hh.value, ww.value, 3), np.uint8)
        self._check(self._lib.dfd_memcpy_d2h(self._p, _ptr(out), self.frame_ptr(), out.nbytes))
        return out

    def decode_jpeg_batch(self, datas) -> np.ndarray:
        """n JPEGs of one size -> (n,H,W,3) uint8 BGR through the batch path (device entropy decoding where it applies)"""
        n = len(datas)
        bufs = [(C.c_char * len(d)).from_buffer_copy(d) for d in datas]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_size_t * n)(*[len(d) for d in datas])
        hh, ww = C.c_int(), C.c_int()
        self._check(self._lib.dfd_decode_jpeg_batch(self._p, n, ptrs, lens, None, 0, C.byref(hh), C.byref(ww)))
        out = np.empty((n, hh.value, ww.value, 3), np.uint8)
        self._check(self._lib.dfd_memcpy_d2h(self._p, _ptr(out), self.frame_ptr(), out.nbytes))
        return out

    def jpeg_decode_counts(self):
        """(frames of batch calls entropy-decoded on the device, frames decoded by the host decoder)"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        self._check(self._lib.dfd_jpeg_decode_counts(self._p, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def frame_ptr(self) -> int:
        """device address of the last uploaded / decoded frame"""
        return self._lib.dfd_frame_ptr(self._p)

    def analyze_jpeg(self, data: bytes, full_forensics: bool, stream_id: int = 0, confidence_threshold: float = 0.5,
                     max_faces: int = 16, apply_clahe: bool = True):
        """analyze_frame from JPEG bytes, decoded on the device -> (scores, forensic prob, boxes, logits, (H, W))"""
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        max_faces = max(1, int(max_faces))
        sc = np.empty(6, np.float64)
        prob = C.c_double()
        boxes = np.zeros((max_faces, 4), np.int32)
        logits = np.zeros(max_faces, np.float32)
        n, hh, ww = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.dfd_analyze_jpeg(self._p, int(stream_id), buf, len(data), int(bool(full_forensics)),
                                               float(confidence_threshold), max_faces, int(bool(apply_clahe)), _ptr(sc),
                                               C.byref(prob), _ptr(boxes), C.byref(n), _ptr(logits), C.byref(hh), C.byref(ww)))
        scores = {k: float(v) for k, v in zip(self.FORENSIC_KEYS, sc) if not np.isnan(v)}
        return (scores, float(prob.value), [tuple(int(v) for v in boxes[i]) for i in range(n.value)], logits[: n.value].copy(),
                (hh.value, ww.value))

    def analyze_stream_batch(self, items, full_flags, stream_id: int = 0, confidence_threshold: float = 0.5,
                             max_faces: int = 1, apply_clahe: bool = True):
        """n consecutive frames of one stream in ONE call (dfd_analyze_stream_batch).  items: JPEG `bytes` and / or BGR
        uint8 arrays of one size.  -> list of (scores dict, forensic prob, boxes, logits, n_detected) per frame, (H, W)"""
        n = len(items)
        if n == 0 or len(full_flags) != n:
            raise ValueError("analyze_stream_batch: one full / fast flag per frame")
        keep, ptrs, lens = [], (C.c_void_p * n)(), (C.c_size_t * n)()
        hh = ww = 0
        for i, it in enumerate(items):
            if isinstance(it, (bytes, bytearray, memoryview)):
                b = (C.c_char * len(it)).from_buffer_copy(it)
                keep.append(b)
                ptrs[i], lens[i] = C.addressof(b), len(it)
            else:
                a = np.ascontiguousarray(it, dtype=np.uint8)
                if a.ndim != 3 or a.shape[2] != 3 or (hh and a.shape[:2] != (hh, ww)):
                    raise ValueError("analyze_stream_batch: raw frames are (H, W, 3) uint8 of one size")
                hh, ww = a.shape[:2]
                keep.append(a)
                ptrs[i], lens[i] = a.ctypes.data, 0
        max_faces = max(1, int(max_faces))
        full = np.ascontiguousarray([int(bool(f)) for f in full_flags], dtype=np.int32)
        sc = np.empty((n, 6), np.float64)
        prob = np.empty(n, np.float64)
        boxes = np.zeros((n, max_faces, 4), np.int32)
        nf = np.zeros(n, np.int32)
        nd = np.zeros(n, np.int32)
        logits = np.zeros((n, max_faces), np.float32)
        oh, ow = C.c_int(), C.c_int()
        self._check(self._lib.dfd_analyze_stream_batch(self._p, int(stream_id), n, ptrs, lens, int(hh), int(ww), _ptr(full),
                                                       float(confidence_threshold), max_faces, int(bool(apply_clahe)), _ptr(sc),
                                                       _ptr(prob), _ptr(boxes), _ptr(nf), _ptr(nd), _ptr(logits),
                                                       C.byref(oh), C.byref(ow)))
        out = []
        for i in range(n):
            scores = {k: float(v) for k, v in zip(self.FORENSIC_KEYS, sc[i]) if not np.isnan(v)}
            out.append((scores, float(prob[i]), [tuple(int(v) for v in boxes[i, j]) for j in range(nf[i])],
                        logits[i, : nf[i]].copy(), int(nd[i])))
        return out, (oh.value, ow.value)

    def host_alloc(self, shape, dtype=np.uint8) -> np.ndarray:
        """A numpy array over pinned host memory (hipHostMalloc); release with host_free(arr)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._lib.dfd_host_alloc(self._p, nbytes, C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self._check(self._lib.dfd_host_free(self._p, p))

    def analyze_frames_host(self, frames: np.ndarray, batch: int, forced_boxes=None, confidence_threshold: float = 0.5,
                            max_faces: int = 4, apply_clahe: bool = True, with_forensics: bool = False):
        """frames: (n, H, W, 3) uint8 in host memory (pinned: host_alloc) -> as analyze_batch_device, PCIe included."""
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3 or not frames.flags["C_CONTIGUOUS"]:
            raise ValueError("expected a contiguous (n,H,W,3) uint8 array")
        n, height, width = frames.shape[:3]
        forced, forced_k = None, 0
        if forced_boxes is not None:
            forced = np.ascontiguousarray(np.asarray(forced_boxes, np.int32).reshape(n, -1, 4))
            forced_k = forced.shape[1]
        xy = np.zeros((n, max_faces, 4), np.int32)
        nf = np.zeros(n, np.int32)
        lg = np.zeros((n, max_faces), np.float32)
        fp = np.zeros(n, np.float64)
        self._check(self._lib.dfd_analyze_frames_host(
            self._p, frames.ctypes.data, n, int(batch), height, width, _ptr(forced) if forced is not None else None, forced_k,
            float(confidence_threshold), int(max_faces), int(bool(apply_clahe)), int(bool(with_forensics)),
            _ptr(xy), _ptr(nf), _ptr(lg), _ptr(fp)))
        boxes = [[tuple(int(v) for v in xy[f, i]) for i in range(nf[f])] for f in range(n)]
        return boxes, [lg[f, : nf[f]].copy() for f in range(n)], (fp if with_forensics else None)

    def pack_jpegs(self, datas):
        """the files back to back (64-byte aligned) in ONE pinned buffer -> (buffer, offsets, lengths); release the buffer
        with host_free.  What analyze_jpegs_host takes for full speed: DMA straight from the caller's memory."""
        offs, total = [], 0
        for d in datas:
            offs.append(total)
            total += (len(d) + 63) // 64 * 64
        buf = self.host_alloc((max(total, 64),))
        for o, d in zip(offs, datas):
            buf[o:o + len(d)] = np.frombuffer(d, np.uint8)
        return buf, offs, [len(d) for d in datas]

    def analyze_jpegs_host(self, datas, batch: int, forced_boxes=None, confidence_threshold: float = 0.5, max_faces: int = 4,
                           apply_clahe: bool = True, with_forensics: bool = False, packed=None):
        """JPEG files of one size (bytes objects, or packed = pack_jpegs(...) for pinned input) -> as analyze_batch_device:
        the files' bytes cross PCIe and are entropy-decoded on the device (dfd_analyze_jpegs_host)."""
        n = len(datas) if packed is None else len(packed[1])
        if packed is None:
            keep = [(C.c_char * len(d)).from_buffer_copy(d) for d in datas]
            ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in keep])
            lens = (C.c_size_t * n)(*[len(d) for d in datas])
        else:
            buf, offs, ls = packed
            ptrs = (C.c_void_p * n)(*[buf.ctypes.data + o for o in offs])
            lens = (C.c_size_t * n)(*ls)
        forced, forced_k = None, 0
        if forced_boxes is not None:
            forced = np.ascontiguousarray(np.asarray(forced_boxes, np.int32).reshape(n, -1, 4))
            forced_k = forced.shape[1]
        xy = np.zeros((n, max_faces, 4), np.int32)
        nf = np.zeros(n, np.int32)
        lg = np.zeros((n, max_faces), np.float32)
        fp = np.zeros(n, np.float64)
        hh, ww = C.c_int(), C.c_int()
        self._check(self._lib.dfd_analyze_jpegs_host(
            self._p, ptrs, lens, n, int(batch), _ptr(forced) if forced is not None else None, forced_k, float(confidence_threshold),
            int(max_faces), int(bool(apply_clahe)), int(bool(with_forensics)), _ptr(xy), _ptr(nf), _ptr(lg), _ptr(fp),
            C.byref(hh), C.byref(ww)))
        boxes = [[tuple(int(v) for v in xy[f, i]) for i in range(nf[f])] for f in range(n)]
        return boxes, [lg[f, : nf[f]].copy() for f in range(n)], (fp if with_forensics else None), (hh.value, ww.value)

    def classifier_crop_count(self) -> int:
        """crops the classifier has run on since the handle was created (a crop the MTCNN stage rejects is not one)"""
        v = C.c_ulonglong(0)
        self._check(self._lib.dfd_classifier_crop_count(self._p, C.byref(v)))
        return int(v.value)

    def last_detection_count(self) -> int:
        """len(faces) of the last detect_faces / analyze_frame call, before its max_out / max_faces cut"""
        return int(self._lib.dfd_last_detection_count(self._p))

    def analyze_batch_device(self, frames_dev: int, n: int, height: int, width: int, forced_boxes=None,
                             confidence_threshold: float = 0.5, max_faces: int = 4, apply_clahe: bool = True,
                             with_forensics: bool = False):
        """n frames resident in HBM -> (boxes per frame, logits per frame, forensic probabilities or None)."""
        forced = None
        forced_k = 0
        if forced_boxes is not None:
            forced = np.ascontiguousarray(np.asarray(forced_boxes, np.int32).reshape(n, -1, 4))
            forced_k = forced.shape[1]
        xy = np.zeros((n, max_faces, 4), np.int32)
        nf = np.zeros(n, np.int32)
        lg = np.zeros((n, max_faces), np.float32)
        fp = np.zeros(n, np.float64)
        self._check(self._lib.dfd_analyze_batch_device(
            self._p, frames_dev, int(n), int(height), int(width), _ptr(forced) if forced is not None else None, forced_k,
            float(confidence_threshold), int(max_faces), int(bool(apply_clahe)), int(bool(with_forensics)),
            _ptr(xy), _ptr(nf), _ptr(lg), _ptr(fp)))
        boxes = [[tuple(int(v) for v in xy[f, i]) for i in range(nf[f])] for f in range(n)]
        logits = [lg[f, : nf[f]].copy() for f in range(n)]
        return boxes, logits, (fp if with_forensics else None)

    def forensic_signals_device(self, frames_dev: int, n: int, height: int, width: int, prev_index):
        """-> (scores [n][5] = frequency, noise, ela, edge, color; mean_diff [n], -1 where prev_index < 0).
        prev_index[f] = -2 marks a frame that is only a predecessor (tail of the batch): no signals, outputs -1."""
        pi = np.ascontiguousarray(np.asarray(prev_index, np.int32).reshape(-1))
        if pi.size != n:
            raise ValueError("prev_index must have one entry per frame")
        sc = np.empty((n, 5), np.float64)
        md = np.empty(n, np.float64)
        self._check(self._lib.dfd_forensic_signals_device(self._p, frames_dev, int(n), int(height), int(width), _ptr(pi),
                                                          _ptr(sc), _ptr(md)))
        return sc, md

    # -- vote exchange over RCCL (C ABI)
    COMM_ID_BYTES = 128

    @staticmethod
    def comm_unique_id() -> bytes:
        lib = load()
        buf = (C.c_char * Handle.COMM_ID_BYTES)()
        rc = lib.dfd_comm_unique_id(buf)
        if rc != 0:
            raise DfdError(rc, (lib.dfd_last_error(None) or b"dfd_comm_unique_id failed").decode())
        return bytes(buf)

    def comm_init(self, comm_id: bytes, rank: int, world: int):
        if len(comm_id) != self.COMM_ID_BYTES:
            raise ValueError("communicator id must be 128 bytes")
        buf = (C.c_char * self.COMM_ID_BYTES).from_buffer_copy(comm_id)
        self._check(self._lib.dfd_comm_init(self._p, buf, int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._lib.dfd_comm_destroy(self._p))

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        self._check(self._lib.dfd_comm_info(self._p, C.byref(r), C.byref(w)))
        return r.value, w.value

    def vote_allgather(self, local: np.ndarray) -> np.ndarray:
        """One ncclAllGather of this rank's record block -> (world, *local.shape), rank-major."""
        a = np.ascontiguousarray(local)
        _, world = self.comm_info()
        if world <= 0:
            raise DfdError(-5, "vote_allgather: no communicator (comm_init)")
        out = np.empty((world,) + a.shape, a.dtype)
        self._check(self._lib.dfd_vote_allgather(self._p, _ptr(a), a.nbytes, _ptr(out)))
        return out

    def vote_allgather_waves(self, local: np.ndarray) -> np.ndarray:
        """(waves, *block) of this rank -> (waves, world, *block): one all-gather per wave, one upload / download / wait"""
        a = np.ascontiguousarray(local)
        _, world = self.comm_info()
        if world <= 0:
            raise DfdError(-5, "vote_allgather_waves: no communicator (comm_init)")
        out = np.empty((a.shape[0], world) + a.shape[1:], a.dtype)
        self._check(self._lib.dfd_vote_allgather_waves(self._p, _ptr(a), a.shape[0], a.nbytes // a.shape[0], _ptr(out)))
        return out

    @property
    def has_mtcnn(self) -> bool:
        return bool(self._lib.dfd_has_mtcnn(self._p))

    def mtcnn_align(self, face_bgr):
        """MTCNN.forward on an already-cropped face (reference deepfake_detection.py:376-377):
        ((3,160,160) float32 RGB 0..255, (x1,y1,x2,y2,prob)) or (None, None) when no face passes."""
        a = self._as_bgr(face_bgr)
        face = np.empty((3, 160, 160), np.float32)
        box = np.empty(5, np.float32)
        found = C.c_int()
        self._check(self._lib.dfd_mtcnn_align(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], _ptr(face),
                                              _ptr(box), C.byref(found)))
        return (face, box) if found.value else (None, None)

    def mtcnn_tap(self, face_bgr, name: str, capacity: int = 1 << 20) -> np.ndarray:
        a = self._as_bgr(face_bgr)
        out = np.empty(int(capacity), np.float32)
        cnt = C.c_size_t()
        dims = np.zeros(3, np.int32)
        self._check(self._lib.dfd_mtcnn_tap(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], name.encode(),
                                            _ptr(out), out.size, C.byref(cnt), _ptr(dims)))
        shape = tuple(int(d) for d in dims if d != 1) or (1,)
        if dims[1] == 5 and dims[2] == 1:
            shape = (int(dims[0]), 5)
        return out[: cnt.value].reshape(shape) if cnt.value else out[:0].reshape((0,) + shape[1:])

    def ssd_tap(self, frame, name: str, capacity: int) -> np.ndarray:
        a = self._as_bgr(frame)
        out = np.empty(int(capacity), np.float32)
        cnt = C.c_size_t()
        self._check(self._lib.dfd_ssd_tap(self._p, _ptr(a), a.shape[0], a.shape[1], a.strides[0], name.encode(),
                                          _ptr(out), out.size, C.byref(cnt)))
        return out[: cnt.value]


class ClassifierLanes:
    """Several classifier forwards in flight on ONE device: `lanes` handles (each with its own stream and workspace,
    the same weights and the same measured GEMM tiles) take batches in turn, every call asynchronous.

    One batch-256 forward is ~54 dependent launches whose late layers (14 x 14 / 7 x 7 maps) leave CUs idle and whose
    launch gaps are 4-5 us each; a second forward's kernels fill both (2.97 -> 2.78 ms per
    batch).  The reference serves one frame at a time (deepfake_detection.py:357-406); a server that has two batches
    queued is the case this class is for, and the results are the same bits whichever lane computes a batch."""

    def __init__(self, blob: bytes, device: int = 0, max_batch: int = 8, lanes: int = 2, first: "Handle" = None):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.handles = [first if first is not None else Handle(blob, device=device, max_batch=max_batch)]
        for _ in range(lanes - 1):
            self.handles.append(Handle(blob, device=device, max_batch=max_batch))
        self._owned = self.handles[1:] if first is not None else list(self.handles)
        self._next = 0
        # (if the lanes do not overlap in a process that has made many streams - two main streams mapped onto one hardware
        # queue run in line - give one lane another priority: handles[1].set_option("stream_priority", 1))

    def __len__(self):
        return len(self.handles)

    def warmup(self, n_crops: int):
        """lane 0 measures the tiles, the others take its table (a lane that tuned for itself could pick another of the
        bit-identical tiles; the table is shared so that every lane launches the same kernels)"""
        self.handles[0].warmup(n_crops, 0)
        table = self.handles[0].tiles_export()
        for h in self.handles[1:]:
            h.tiles_import(table)
            h.warmup(n_crops, 0)

    def set_option(self, name: str, value: int):
        for h in self.handles:
            h.set_option(name, value)

    def submit(self, x_dev: int, n: int, logits_dev: int) -> int:
        """queue one forward on the next lane (asynchronous); -> the lane index it went to"""
        k = self._next
        self.handles[k].classify_device(x_dev, n, logits_dev)
        self._next = (k + 1) % len(self.handles)
        return k

    def check_overlap(self, x_dev: int, n: int, outs, steps: int = 6, gain: float = 0.97):
        """Untimed check that the lanes really run side by side (the runtime may have mapped their main streams onto one
        hardware queue - then they run in line and two lanes are no faster than one).  Times `steps`
        forwards on lane 0 alone and on all lanes; if the lanes are not at least 3 % faster, lane 1's main stream is
        re-created in the high-priority pool and the measurement repeated, and whichever setting was faster stays.
        -> {"one_lane_ms", "lanes_ms", "lane1_priority", ...} (per forward)."""
        import time

        def run(handles):
            for h in handles:
                h.sync()
            t0 = time.perf_counter()
            for i in range(steps):
                handles[i % len(handles)].classify_device(x_dev, n, outs[i % len(handles)].ptr)
            for h in handles:
                h.sync()
            return (time.perf_counter() - t0) / steps * 1e3

        run(self.handles)                                           # clocks, first-use costs
        rep = {"one_lane_ms": round(run(self.handles[:1]), 4), "lanes_ms": round(run(self.handles), 4), "lane1_priority": 0}
        if len(self.handles) > 1 and rep["lanes_ms"] > gain * rep["one_lane_ms"]:
            self.handles[1].set_option("stream_priority", 1)
            rep["lanes_ms_high_priority"] = round(run(self.handles), 4)
            if rep["lanes_ms_high_priority"] < rep["lanes_ms"]:
                rep["lane1_priority"] = 1
            else:
                self.handles[1].set_option("stream_priority", 0)
        self._next = 0
        return rep

    def submit_alone(self, x_dev: int, n: int, logits_dev: int) -> int:
        """queue one forward that runs with NO other lane's kernels beside it, ordered on the device (dfd_wait_for: no
        host wait, the host stays ahead of the GPU): it starts when the other lanes have finished what they hold, and they
        continue when it has finished.  bench.py's instrumented step - per-launch events then time isolated kernels."""
        k = self._next
        me = self.handles[k]
        for o in self.handles:
            if o is not me:
                me.wait_for(o)
        me.classify_device(x_dev, n, logits_dev)
        for o in self.handles:
            if o is not me:
                o.wait_for(me)
        self._next = (k + 1) % len(self.handles)
        return k

    def sync(self):
        for h in self.handles:
            h.sync()

    def close(self):
        """closes the handles this object made; a `first` handle passed in stays open and remains the only lane"""
        for h in self._owned:
            h.close()
        self.handles = [h for h in self.handles if not any(h is o for o in self._owned)]
        self._owned = []
        self._next = 0
