"""MI355X-native per-frame deepfake inference path (host side).

Mirrors the reference's Python surface (model.py, face_detection.py,
frame_analysis.py, deepfake_detection.py, backend_server.py) over the C ABI in
include/dfd_hip.h.  There is no CPU fallback: anything that computes needs
libdfd_hip.so and a gfx950 device and raises otherwise.
"""
__all__ = ["b0_arch", "weights"]
