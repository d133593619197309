"""Probe: host time of one asynchronous batch-256 forward call (54 launches through the C ABI) in bench.py's two-lane
loop - is the submitting thread far enough ahead of the GPU for both lanes to stay busy?  Prints the per-call host
times and the loop's wall time per forward."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd  # noqa: E402

B = 256
blob = rtdfd_amd.weights.pack_all(rtdfd_amd.weights.seeded_state_dict(0), rtdfd_amd.weights.seeded_ssd_state_dict(0))
h = rtdfd_amd._lib.Handle(blob, device=0, max_batch=B)
lanes = rtdfd_amd._lib.ClassifierLanes(blob, device=0, max_batch=B, lanes=int(os.environ.get("LANES", "2")), first=h)
lanes.warmup(B)
x = np.random.default_rng(1).standard_normal((B, 3, 224, 224)).astype(np.float32)
xd = h.alloc(x.nbytes).upload(x)
ys = [h.alloc(B * 4) for _ in range(len(lanes))]
for i in range(6):
    lanes.submit(xd.ptr, B, ys[i % len(lanes)].ptr)
lanes.sync()
for rep in range(3):
    call = []
    t0 = time.perf_counter()
    for i in range(40):
        t = time.perf_counter()
        lanes.submit(xd.ptr, B, ys[i % len(lanes)].ptr)
        call.append(time.perf_counter() - t)
    t_enq = time.perf_counter() - t0
    lanes.sync()
    wall = time.perf_counter() - t0
    call = np.array(call) * 1e3
    print(f"lanes={len(lanes)}: wall {wall / 40 * 1e3:.3f} ms per forward; host enqueue total {t_enq / 40 * 1e3:.3f} ms per forward "
          f"(per call: median {np.median(call):.3f}, p90 {np.percentile(call, 90):.3f}, max {call.max():.3f} ms); "
          f"first 8 calls: {np.round(call[:8], 3).tolist()}", flush=True)
