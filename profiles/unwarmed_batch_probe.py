import sys, time, numpy as np
sys.path.insert(0, ".")
import rtdfd_amd
W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_b0(W.seeded_state_dict(0)), device=0, max_batch=256)
x = np.random.RandomState(1).randn(128, 3, 224, 224).astype(np.float32)
xd = h.alloc(x.nbytes).upload(x); yd = h.alloc(1024)
out = []
for n in (128, 71, 16, 4, 1):
    for _ in range(5): h.classify_device(xd.ptr, n, yd.ptr)
    h.sync(); t0 = time.perf_counter()
    for _ in range(40): h.classify_device(xd.ptr, n, yd.ptr)
    h.sync(); out.append((n, round((time.perf_counter() - t0) / 40 * 1e3, 3)))
print(out)
