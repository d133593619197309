"""Times the e2e batch path in both orders, to separate a real difference from warm-up/clock effects."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd
W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0)), device=0, max_batch=256)
N, H, Wd, K = 64, 1080, 1920, 4
frames = np.random.default_rng(7).integers(50, 200, (N, H, Wd, 3), dtype=np.uint8)
fd = h.alloc(frames.nbytes).upload(frames)
boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * N
def run(forensic, steps=10):
    h.sync(); t0 = time.perf_counter()
    for _ in range(steps):
        h.analyze_batch_device(fd.ptr, N, H, Wd, forced_boxes=boxes, max_faces=K, with_forensics=forensic)
    h.sync(); return N * steps / (time.perf_counter() - t0)
for f in (False, True): run(f, 3)
for order in ((False, True), (True, False), (False, True)):
    print({("forensics" if f else "plain"): round(run(f)) for f in order})
