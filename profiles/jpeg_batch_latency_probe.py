"""dfd_decode_jpeg_batch: device entropy decoder against the host pool, per batch shape (what /analyze_batch sees)."""
import io, os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from PIL import Image
import rtdfd_amd
import frames as F

W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_b0(W.seeded_state_dict(0)), device=0, max_batch=8)
def enc(fr):
    b = io.BytesIO(); Image.fromarray(np.ascontiguousarray(fr[..., ::-1])).save(b, format="JPEG", quality=85); return b.getvalue()
for (n, hh, ww) in ((2, 480, 640), (8, 480, 640), (32, 480, 640), (2, 1080, 1920), (8, 1080, 1920), (32, 1080, 1920)):
    files = [enc(F.natural_like(hh, ww, seed=3 + i)) for i in range(min(n, 4))]
    datas = [files[i % len(files)] for i in range(n)]
    bufs = [(C.c_char * len(d)).from_buffer_copy(d) for d in datas]
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs]); lens = (C.c_size_t * n)(*[len(d) for d in datas])
    a, b = C.c_int(), C.c_int()
    out = {}
    for mode in (0, 1):
        h.set_option("jpeg_device_entropy", mode)
        for _ in range(3): h._lib.dfd_decode_jpeg_batch(h._p, n, ptrs, lens, None, 0, C.byref(a), C.byref(b))
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); h._lib.dfd_decode_jpeg_batch(h._p, n, ptrs, lens, None, 0, C.byref(a), C.byref(b)); ts.append(time.perf_counter() - t0)
        out[mode] = sorted(ts)[3] * 1e3
    print(f"{n:3d} x {hh}x{ww} ({len(files[0])} B each): host pool {out[0]:.2f} ms, device {out[1]:.2f} ms", flush=True)
