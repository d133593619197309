"""Times every fused expand+depthwise variant (csrc/b0_kernels.hip DFD_MB2_TABLE) at batch 256 and checks each
against the CPU oracle on the first 8 crops.  `python profiles/mb_variants.py [bf16]`"""
import json
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
torch.manual_seed(1)
x = torch.randn(256, 3, 224, 224)
want = b0_ref.forward(W.to_torch(sd), x[:8]).numpy()
xn = x.numpy()
xd = h.alloc(xn.nbytes).upload(xn)
yd = h.alloc(1024)
h.warmup(256, 0)
blocks = {(112, 2): "b1.dw", (56, 1): "b2.dw", (56, 2): "b3.dw", (28, 1): "b4.dw", (28, 2): "b5.dw"}
# -1 = first generation (fp32, stride 2 only), 0-1 = DFD_MB2_TABLE, 6.. = DFD_MB3_TABLE (round 3: NT / INS)
nvar = {(112, 2): [-1] if not bf16 else [0, 1], (56, 1): [6, 18], (56, 2): [-1], (28, 1): [13, 18],
        (28, 2): [-3]}
if os.environ.get("MB_VARS"):
    nvar = {k: [int(v) for v in os.environ["MB_VARS"].split(",")] for k in nvar}


def run():
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
    return {n: ms / steps for n, ms in layers}, float(np.abs(y[:8] - want).max())


base, err = run()
print("baseline", {k: round(base[v] * 1e3, 1) for k, v in blocks.items()}, "us; err", err, flush=True)
for key, layer in blocks.items():
    for v in nvar[key]:
        os.environ[f"DFD_MB_VARIANT_{key[0]}_{key[1]}"] = str(v)
        t, err = run()
        print(f"{layer} variant {v}: {t[layer] * 1e3:7.1f} us   (step dw total {sum(ms for n, ms in t.items() if n.endswith('.dw')) * 1e3:.0f} us)  err {err:.2e}", flush=True)
    del os.environ[f"DFD_MB_VARIANT_{key[0]}_{key[1]}"]
