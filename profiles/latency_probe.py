"""Single-request latencies of the server flow (one frame per call): classifier batch 1 / 4, detector, full analyze_frame."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd as pkg
W = pkg.weights
h = pkg._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0)), device=0, max_batch=16)
for n in (1, 4, 16):
    h.warmup(n, 1 if n == 1 else 0)
rng = np.random.default_rng(7)
frame = rng.integers(50, 200, (1080, 1920, 3), dtype=np.uint8)
x = rng.standard_normal((16, 3, 224, 224)).astype(np.float32)

def t_ms(fn, reps=50):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3

xd = h.alloc(x.nbytes).upload(x)
yd = h.alloc(16 * 4)
for n in (1, 4, 16):
    def run():
        h.classify_device(xd.ptr, n, yd.ptr)
        h.sync()
    print(f"classify_device batch {n}: {t_ms(run):.3f} ms")
print(f"detect_faces 1080p (upload + detector): {t_ms(lambda: h.detect_faces(frame)):.3f} ms")
print(f"analyze_frame full forensics, max_faces 1: {t_ms(lambda: h.analyze_frame(frame, True, stream_id=5, max_faces=1)):.3f} ms")
print(f"analyze_frame fast forensics, max_faces 1: {t_ms(lambda: h.analyze_frame(frame, False, stream_id=5, max_faces=1)):.3f} ms")
