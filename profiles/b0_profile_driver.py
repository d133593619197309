"""A few batch-256 classifier forwards and nothing else, for rocprofv3 (kernel trace or --pmc passes).

    DFD_TILE_CACHE=gpurun_out/tiles.txt python profiles/b0_profile_driver.py          # un-profiled: measures + saves tiles
    rocprofv3 --pmc ... -- python3 profiles/b0_profile_driver.py                       # profiled: loads them, no tuning launches

env: B0_STEPS (default 4), B0_BF16=1 (bf16 activation storage), DFD_MB_VARIANT_* as usual."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rtdfd_amd  # noqa: E402

W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_b0(W.seeded_state_dict(0)), device=0, max_batch=256)
h.set_option("bf16_activations", int(os.environ.get("B0_BF16", "0")))
x = np.random.RandomState(1).randn(256, 3, 224, 224).astype(np.float32)
xd = h.alloc(x.nbytes).upload(x)
yd = h.alloc(1024)
h.warmup(256, 0)
for _ in range(int(os.environ.get("B0_STEPS", "4"))):
    h.classify_device(xd.ptr, 256, yd.ptr)
h.sync()
print("measured tile entries:", len(h.tiles_export().splitlines()), "forwards:", 1 + int(os.environ.get("B0_STEPS", "4")))
h.close()
