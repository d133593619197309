"""Does running two independent half-batches on two HIP streams beat one full batch?  (two handles, two host threads)"""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
import rtdfd_amd
W = rtdfd_amd.weights
blob = W.pack_b0(W.seeded_state_dict(0))
x = np.random.RandomState(1).randn(256, 3, 224, 224).astype(np.float32)

def bench(handles, n_each, steps=30):
    bufs = []
    for h in handles:
        h.warmup(n_each, 0)
        xd = h.alloc(x[:n_each].nbytes).upload(x[:n_each]); yd = h.alloc(1024)
        bufs.append((xd, yd))
    def loop(h, xd, yd, k):
        for _ in range(k):
            h.classify_device(xd.ptr, n_each, yd.ptr)
        h.sync()
    for (h, (xd, yd)) in zip(handles, bufs): loop(h, xd, yd, 5)
    ts = [threading.Thread(target=loop, args=(h, xd, yd, steps)) for h, (xd, yd) in zip(handles, bufs)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    return n_each * len(handles) * steps / dt

h1 = rtdfd_amd._lib.Handle(blob, device=0, max_batch=256)
print("1 stream x 256:", round(bench([h1], 256)), "crops/s", flush=True)
h2 = rtdfd_amd._lib.Handle(blob, device=0, max_batch=256)
print("2 streams x 128:", round(bench([h1, h2], 128)), "crops/s", flush=True)
print("2 streams x 256:", round(bench([h1, h2], 256)), "crops/s", flush=True)
h3 = rtdfd_amd._lib.Handle(blob, device=0, max_batch=128); h4 = rtdfd_amd._lib.Handle(blob, device=0, max_batch=128)
print("4 streams x 64:", round(bench([h1, h2, h3, h4], 64)), "crops/s", flush=True)
for bf in (1,):
    for h in (h1, h2): h.set_option("bf16_activations", 1)
    print("bf16 1 stream x 256:", round(bench([h1], 256)), flush=True)
    print("bf16 2 streams x 128:", round(bench([h1, h2], 128)), flush=True)
