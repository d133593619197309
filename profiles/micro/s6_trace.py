"""Cycle trace of one wave of the split GEMM (build csrc with `make EXTRA=-DS6_TRACE` first).

    python profiles/micro/s6_trace.py            # batch 256 forward; trace of the K=1152, N=192 projection

Prints, per stage of block 8 / wave 0: cycles from stage start to (2) loads issued, (3) MFMAs issued,
(4) LDS stores issued, (5) barrier passed.  s_memtime ticks are shader-clock cycles on gfx950."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rtdfd_amd  # noqa: E402

W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_all(W.seeded_state_dict(0), None), device=0, max_batch=256)
x = np.random.RandomState(0).randn(256, 3, 224, 224).astype(np.float32)
xd = h.alloc(x.nbytes).upload(x)
yd = h.alloc(256 * 4)
for _ in range(3):
    h.classify_device(xd.ptr, 256, yd.ptr)
h.sync()
lib = h._lib
out = (C.c_longlong * 1024)()
lib.dfd_debug_s6_trace.argtypes = [C.c_void_p, C.c_int]
lib.dfd_debug_s6_trace.restype = C.c_int
assert lib.dfd_debug_s6_trace(out, 1024) == 0
n = int(out[1023])
ev = [(int(out[i]), int(out[i + 1])) for i in range(0, n, 2)]
t0 = ev[0][1]
print("events", len(ev), "total cycles", ev[-1][1] - t0)
prev = t0
row = []
for ident, t in ev:
    if ident == 1 and row:
        print(" ".join(row))
        row = []
    row.append(f"{ident}:+{t - prev}")
    prev = t
print(" ".join(row))
