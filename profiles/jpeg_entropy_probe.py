"""Host entropy decoder alone (no GPU work): wall time of dfd_jpeg_coefficients on the bench's 1080p request body for
the pool size given by DFD_HOST_THREADS; DFD_JPEG_VERBOSE=1 prints the phases."""
import ctypes as C
import io
import os
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtdfd_amd as pkg  # noqa: E402

rs = np.random.default_rng(7)
fr = rs.integers(50, 200, (1080, 1920, 3), dtype=np.uint8)
buf = io.BytesIO()
Image.fromarray(fr[..., ::-1]).save(buf, "JPEG", quality=85)
data = buf.getvalue()
lib = pkg._lib.load()
b = (C.c_char * len(data)).from_buffer_copy(data)
info = (C.c_int * 16)()
cnt = C.c_size_t()
coef = np.zeros(4_000_000, np.int16)
q = np.zeros((4, 64), np.uint16)
ts = []
for i in range(12):
    t = time.perf_counter()
    rc = lib.dfd_jpeg_coefficients(b, len(data), info, q.ctypes.data_as(C.c_void_p), coef.ctypes.data_as(C.c_void_p), coef.size, C.byref(cnt))
    ts.append((time.perf_counter() - t) * 1e3)
    assert rc == 0
print(f"threads {os.environ.get('DFD_HOST_THREADS', 'default')}: jpeg {len(data)} bytes, decode ms min {min(ts):.2f} median {sorted(ts)[len(ts) // 2]:.2f}; cpus {os.cpu_count()} affinity {len(os.sched_getaffinity(0))}")
