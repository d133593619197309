"""bf16-activation logit error statistics against the fp32 HIP path (itself within 1e-5 of the CPU oracle):
which kernel choices (fused stem on the MFMA, fused expand) move them, and how large the error is relative to the
logit spread.  Feeds the tolerance in tests/test_b0_bf16_gpu.py and DESIGN section 4a.
  python profiles/bf16_logit_stats.py > gpurun_out/bf16_logit_stats.txt"""
import sys
sys.path.insert(0, '.')
import numpy as np
import rtdfd_amd
W = rtdfd_amd.weights
sd = W.seeded_state_dict(0)
h = rtdfd_amd._lib.Handle(W.pack_b0(sd), device=0, max_batch=64)


def crops(n, seed):
    rs = np.random.RandomState(seed)
    return (rs.randn(n, 3, 224, 224) * np.linspace(0.4, 1.8, n).reshape(n, 1, 1, 1)).astype(np.float32)


for seed, n in ((21, 3), (21, 64), (5, 64), (77, 64)):
    x = crops(n, seed)
    h.set_option("bf16_activations", 0)
    ref = h.classify(x).ravel()
    for fs in (0, 1):
        for fe in (0, 1):
            h.set_option("bf16_activations", 1)
            h.set_option("fuse_stem", fs)
            h.set_option("fuse_expand", fe)
            y = h.classify(x).ravel()
            d = np.abs(y - ref)
            print(f"seed {seed} n {n} fuse_stem {fs} fuse_expand {fe}: max {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e} "
                  f"p95 {np.quantile(d, 0.95):.3e} | logits: std {ref.std():.3f} range [{ref.min():.3f}, {ref.max():.3f}] "
                  f"max|d|/std {d.max() / ref.std():.3e}", flush=True)
    h.set_option("fuse_stem", 1)
    h.set_option("fuse_expand", 1)
