"""A/B of the e2e rows that carry the forensic signals: overlap on (second stream) / off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import rtdfd_amd as pkg
W = pkg.weights
NF = 64
sel = W.seeded_mtcnn_state_dict(0, W.MTCNN_SELECTIVE)
h = pkg._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0), sel), device=0, max_batch=4 * NF)
frames = np.random.default_rng(7).integers(50, 200, (NF, 1080, 1920, 3), dtype=np.uint8)
boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * NF
h.warmup(4 * NF, NF)
fd = h.alloc(frames.nbytes).upload(frames)
for mt in (1, 0):
    h.set_option("mtcnn", mt)
    for wf, ov in ((False, 1), (True, 1), (True, 0), (True, 1), (True, 0)):
        h.set_option("overlap_forensics", ov)
        for _ in range(3):
            h.analyze_batch_device(fd.ptr, NF, 1080, 1920, forced_boxes=boxes, max_faces=4, with_forensics=wf)
        ts = []
        for _ in range(11):
            t0 = time.perf_counter()
            h.analyze_batch_device(fd.ptr, NF, 1080, 1920, forced_boxes=boxes, max_faces=4, with_forensics=wf)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"mtcnn {mt} forensics {wf} overlap {ov}: median {ts[5]*1e3:.3f} ms  min {ts[0]*1e3:.3f}  -> {NF/ts[5]:.0f} frames/s")
