"""Device entropy decoder (csrc/jpeg_gpu_entropy.h): time of dfd_decode_jpeg_batch for 64 x 1080p frames and the lanes
that decode per round (DFD_JPEG_VERBOSE=1 prints them).  env: JP_N (frames, default 64), JP_KIND (noise | natural)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from PIL import Image
import rtdfd_amd
import frames as F

n = int(os.environ.get("JP_N", "64"))
kind = os.environ.get("JP_KIND", "noise")
W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_b0(W.seeded_state_dict(0)), device=0, max_batch=8)
datas = []
for i in range(min(n, 8)):
    fr = np.random.default_rng(7 + i).integers(50, 200, (1080, 1920, 3), dtype=np.uint8) if kind == "noise" else F.natural_like(1080, 1920, seed=9 + i)
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(fr[..., ::-1])).save(buf, format="JPEG", quality=85)
    datas.append(buf.getvalue())
datas = [datas[i % len(datas)] for i in range(n)]
print(kind, "frames", n, "bytes per frame", [len(d) for d in datas[:4]], flush=True)
for chunk in [int(c) for c in os.environ.get("JP_CHUNKS", "512,1024,2048").split(",")]:
    h.set_option("jpeg_chunk_bytes", chunk)
    for rounds in [int(c) for c in os.environ.get("JP_ROUNDS", "16,8").split(",")]:
        h.set_option("jpeg_rounds", rounds)
        h.decode_jpeg_batch(datas[:2])
        import ctypes as C
        bufs = [(C.c_char * len(d)).from_buffer_copy(d) for d in datas]
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        lens = (C.c_size_t * n)(*[len(d) for d in datas])
        hh, ww = C.c_int(), C.c_int()
        lib = h._lib
        lib.dfd_decode_jpeg_batch(h._p, n, ptrs, lens, None, 0, C.byref(hh), C.byref(ww))
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            rc = lib.dfd_decode_jpeg_batch(h._p, n, ptrs, lens, None, 0, C.byref(hh), C.byref(ww))
            ts.append(time.perf_counter() - t0)
        print(f"chunk {chunk} rounds {rounds}: rc {rc} median {sorted(ts)[2]*1e3:.2f} ms per {n} frames = {n/sorted(ts)[2]:.0f} frames/s; counts {h.jpeg_decode_counts()}", flush=True)
