"""Probe: the 1080p resident e2e call (bench `e2e.detect_classify*`) with one handle against two handles driven by two
host threads (ctypes releases the GIL inside dfd_analyze_batch_device; each handle has its own streams and workspaces).
Usage: python profiles/e2e_lanes_probe.py [forensics 0|1]"""
import os
import sys
import threading
import time

import numpy as np

if int(os.environ.get("PROBE_TORCH", "0")):       # as bench.py: torch first, so that the process runs on the HIP runtime
    import torch  # noqa: F401                     # bundled with the wheel (7.0) instead of /opt/rocm's (7.2)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd  # noqa: E402

N, H, W, K = 64, 1080, 1920, 4
forensic = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
blob = rtdfd_amd.weights.pack_all(rtdfd_amd.weights.seeded_state_dict(0), rtdfd_amd.weights.seeded_ssd_state_dict(0))
extra = []
if int(os.environ.get("PROBE_EXTRA", "0")):      # handles created (and used) BEFORE the two under test, as in bench.py: do the
    for _ in range(int(os.environ["PROBE_EXTRA"])):  # later streams still get hardware queues of their own?
        e = rtdfd_amd._lib.Handle(blob, device=0, max_batch=256)
        e.warmup(256, N)
        extra.append(e)
hs = ([extra[0]] if int(os.environ.get("PROBE_REUSE", "0")) else []) + [rtdfd_amd._lib.Handle(blob, device=0, max_batch=256) for _ in range(2)]
hs = hs[:2]
hs[0].warmup(256, N)
hs[1].tiles_import(hs[0].tiles_export())
hs[1].warmup(256, N)
frames = np.random.default_rng(7).integers(50, 200, (N, H, W, 3), dtype=np.uint8)
fd = hs[0].alloc(frames.nbytes).upload(frames)
boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * N


def loop(h, k, out=None):
    for _ in range(k):
        r = h.analyze_batch_device(fd.ptr, N, H, W, forced_boxes=boxes, max_faces=K, with_forensics=forensic)
    if out is not None:
        out.append(r)


# what bench.py has done on its first handle before it reaches the two-call rows (PROBE_PRELUDE = letters):
#   A a second classifier handle (ClassifierLanes) and a few batch-256 forwards on both   B forensic calls on handle 0
#   C the bf16 rows (option on, warm-up, calls, option off)   D handle 1 is created only now (as bench.py's h2)
pre = os.environ.get("PROBE_PRELUDE", "")
if "D" in pre:
    hs[1].close()
if "A" in pre:
    lanes = rtdfd_amd._lib.ClassifierLanes(blob, device=0, max_batch=256, lanes=2, first=hs[0])
    lanes.warmup(256)
    x = np.random.default_rng(1).standard_normal((256, 3, 224, 224)).astype(np.float32)
    xd, y0, y1 = hs[0].alloc(x.nbytes).upload(x), hs[0].alloc(1024), hs[0].alloc(1024)
    for i in range(10):
        lanes.submit(xd.ptr, 256, (y0 if i % 2 == 0 else y1).ptr)
    lanes.sync()
if "B" in pre:
    for _ in range(3):
        hs[0].analyze_batch_device(fd.ptr, N, H, W, forced_boxes=boxes, max_faces=K, with_forensics=True)
if "C" in pre:
    hs[0].set_option("bf16_activations", 1)
    hs[0].warmup(256, 0)
    for _ in range(3):
        hs[0].analyze_batch_device(fd.ptr, N, H, W, forced_boxes=boxes, max_faces=K, with_forensics=True)
    hs[0].set_option("bf16_activations", 0)
if "D" in pre:
    hs[1] = rtdfd_amd._lib.Handle(blob, device=0, max_batch=256)
    hs[1].tiles_import(hs[0].tiles_export())
    hs[1].warmup(256, N)
if int(os.environ.get("PROBE_PRIO", "0")):          # handle 1 main stream from another priority pool (1 high, -1 low)
    hs[1].set_option("stream_priority", int(os.environ["PROBE_PRIO"]))
for h in hs:
    loop(h, 3)
ref = []
loop(hs[0], 1, ref)
for rep in range(3):
    t0 = time.perf_counter()
    loop(hs[0], 12)
    one = N * 12 / (time.perf_counter() - t0)
    outs = [[], []]
    th = [threading.Thread(target=loop, args=(hs[i], 6, outs[i])) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    two = N * 12 / (time.perf_counter() - t0)
    same = all(np.array_equal(np.asarray(o[0][1], np.float32), np.asarray(ref[0][1], np.float32), equal_nan=True) for o in outs)
    print(f"forensics={int(forensic)} one handle {one:8.1f} frames/s   two handles / two threads {two:8.1f} frames/s   logits equal: {same}", flush=True)
