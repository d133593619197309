"""Cycle trace of one thread of the fused expand+depthwise kernel (build csrc with `make EXTRA=-DMB_TRACE`;
-DMB_TRACE_H=<input size> selects the block: 112 = b1, 56 = b2/b3, 28 = b4/b5; the last launch at that size wins).
Points per channel chunk: 1 chunk start, 2 after the barrier, 3 weights requested, 4 expand MFMAs + swish + LDS
tile written, 5 after the barrier, 6 depthwise + pool partials + stores done."""
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
out = (C.c_longlong * 256)()
h._lib.dfd_debug_mb_trace.argtypes = [C.c_void_p, C.c_int]
h._lib.dfd_debug_mb_trace.restype = C.c_int
assert h._lib.dfd_debug_mb_trace(out, 256) == 0
n = int(out[255])
ev = [(int(out[i]), int(out[i + 1])) for i in range(0, n, 2)]
print("total cycles", ev[-1][1] - ev[0][1])
prev = ev[0][1]
print(" ".join(f"{i}:+{t - prev0}" for (i, t), prev0 in zip(ev, [ev[0][1]] + [e[1] for e in ev[:-1]])))
