"""Per-launch times of one classifier step at batch 256 (HIP events on the library's stream, every step instrumented)
and the logit error of the first 8 crops against the CPU oracle.  `python profiles/layers.py [bf16] [planes]`;
environment switches of the kernels (DFD_MB_VARIANT_*, DFD_DW_ROWS7, ...) apply."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtdfd_amd  # noqa: E402
from oracle import b0_ref  # noqa: E402

bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"
W = rtdfd_amd.weights
sd = W.seeded_state_dict(0)
h = rtdfd_amd._lib.Handle(W.pack_b0(sd), device=0, max_batch=256)
h.set_option("bf16_activations", int(bf16))
if len(sys.argv) > 2:
    h.set_option("bf16_weight_planes", int(sys.argv[2]))
torch.manual_seed(1)
x = torch.randn(256, 3, 224, 224)
want = b0_ref.forward(W.to_torch(sd), x[:8]).numpy()
xn = x.numpy()
xd = h.alloc(xn.nbytes).upload(xn)
yd = h.alloc(1024)
h.warmup(256, 0)
for _ in range(3):
    h.classify_device(xd.ptr, 256, yd.ptr)
h.sync()
h.set_option("profile_stride", 1)
h.profile_begin()
for _ in range(10):
    h.classify_device(xd.ptr, 256, yd.ptr)
h.sync()
steps, layers = h.profile_end()
y = yd.download((256, 1))
err = float(np.abs(y[:8] - want).max())
kinds = {}
for n, ms in layers:
    kinds[n.split(".")[-1]] = kinds.get(n.split(".")[-1], 0.0) + ms / steps
print(("bf16" if bf16 else "fp32"), "err vs oracle", f"{err:.2e}", "| by kind (us):", {k: round(v * 1e3, 1) for k, v in kinds.items()},
      "| total", round(sum(kinds.values()) * 1e3, 1))
print("dw:", " ".join(f"{n}={ms / steps * 1e3:.1f}" for n, ms in layers if n.endswith(".dw")))
print("proj:", " ".join(f"{n}={ms / steps * 1e3:.1f}" for n, ms in layers if n.endswith(".proj")))
print("exp:", " ".join(f"{n}={ms / steps * 1e3:.1f}" for n, ms in layers if n.endswith(".exp")))
print("other:", " ".join(f"{n}={ms / steps * 1e3:.1f}" for n, ms in layers if n.split(".")[-1] not in ("dw", "proj", "exp", "se")))
