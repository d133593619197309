import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
import rtdfd_amd
from oracle import b0_ref
W = rtdfd_amd.weights
sd = W.seeded_state_dict(0)
h = rtdfd_amd._lib.Handle(W.pack_b0(sd), device=0, max_batch=256)
torch.manual_seed(1)
x = torch.randn(256, 3, 224, 224)
xn = x.numpy()
want = b0_ref.forward(W.to_torch(sd), x[:16]).numpy()
res = {}
xd = h.alloc(xn.nbytes).upload(xn); yd = h.alloc(1024)
for name, opts in (("fp32", {"bf16_activations": 0}), ("bf16_w3", {"bf16_activations": 1, "bf16_weight_planes": 3}),
                   ("bf16_w1", {"bf16_activations": 1, "bf16_weight_planes": 1})):
    for k, v in opts.items():
        h.set_option(k, v)
    h.warmup(256, 0)
    y = h.classify(xn)
    err = float(np.abs(y[:16] - want).max())
    for _ in range(5): h.classify_device(xd.ptr, 256, yd.ptr)
    h.sync()
    h.set_option("profile_stride", 1); h.profile_begin()
    t0 = time.perf_counter()
    for _ in range(20): h.classify_device(xd.ptr, 256, yd.ptr)
    h.sync(); dt = (time.perf_counter() - t0) / 20
    steps, layers = h.profile_end()
    agg = {}
    for nm, ms in layers:
        k = nm.split('.')[-1]; agg[k] = agg.get(k, 0) + ms / steps
    res[name] = {"ms_per_step": dt * 1e3, "err_vs_oracle": err, "by_kind": {k: round(v, 3) for k, v in agg.items()}, "logits": y[:4].ravel().tolist()}
    print(name, json.dumps(res[name]), flush=True)
    res[name]["y"] = y
print("bf16_w3 vs fp32 max", float(np.abs(res["bf16_w3"]["y"] - res["fp32"]["y"]).max()), "logit range", float(np.ptp(res["fp32"]["y"])))
print("bf16_w1 vs fp32 max", float(np.abs(res["bf16_w1"]["y"] - res["fp32"]["y"]).max()))
# tile sweep in bf16 mode
for planes in (3, 1):
    h.set_option("bf16_activations", 1); h.set_option("bf16_weight_planes", planes)
    base = h.classify(xn[:3])
    bad = []
    for i in range(rtdfd_amd._lib.load().dfd_gemm_tile_count()):
        h.set_option("gemm_tile", i)
        if not np.array_equal(h.classify(xn[:3]), base): bad.append(i)
    h.set_option("gemm_tile", -1)
    print("planes", planes, "tiles differing:", bad)
