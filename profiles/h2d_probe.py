"""PCIe-inclusive path (dfd_analyze_frames_host): frames per call x batch size sweep."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd as pkg
W = pkg.weights
h = pkg._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0)), device=0, max_batch=256)
h.warmup(256, 64)
frames = np.random.default_rng(7).integers(50, 200, (64, 1080, 1920, 3), dtype=np.uint8)
boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * 64
for reps in (4, 8):
    pinned = h.host_alloc((reps * 64, 1080, 1920, 3))
    for r in range(reps):
        pinned[r * 64:(r + 1) * 64] = frames
    for batch in (16, 32, 64):
        h.analyze_frames_host(pinned, batch, forced_boxes=boxes * reps, max_faces=4)
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            h.analyze_frames_host(pinned, batch, forced_boxes=boxes * reps, max_faces=4)
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]
        print(f"frames {reps*64} batch {batch}: {reps*64/dt:.0f} frames/s  ({dt*1e3:.1f} ms)", flush=True)
    h.host_free(pinned)
