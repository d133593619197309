"""GPU: every split-GEMM instance the tuner can pick computes the same bits.

gemm_split.hip offers, per shape, pw6 with 4 waves (KS x MT x NT), pw6 with 8 waves (KS x NT) and the pw7 family;
dfd_warmup keeps the fastest by measurement, so which one runs depends on the batch size and the clock.  The
claim that "the choice never changes a result bit" is checked here for EVERY candidate: dfd_set_option(h,
"gemm_tile", i) forces candidate i % len(candidates(shape)) on every 1x1 conv of B0 (plain and squeeze-excite-gated,
with and without residual, swish / none / relu epilogues) and on every k x k implicit-GEMM conv of the detector,
at ragged M (3 crops: M = 3*49 ... 3*12544 is never a multiple of every block height)."""
import numpy as np
import pytest
import torch

import frames as F
from oracle import b0_ref

pytestmark = pytest.mark.gpu


def _crops(n, seed):
    rs = np.random.RandomState(seed)
    return (rs.randn(n, 3, 224, 224) * np.linspace(0.4, 1.8, n).reshape(n, 1, 1, 1)).astype(np.float32)


def test_every_tile_gives_identical_bits_b0_and_detector(pkg, b0_handle, seeded_sd):
    h = b0_handle
    count = pkg._lib.load().dfd_gemm_tile_count()
    x = _crops(3, 5)
    xd = h.alloc(x.nbytes).upload(x)
    frame = F.natural_like(480, 640, seed=11)
    taps = ("b1.out", "b3.out", "b8.out", "b12.out", "head")          # K = 96 (gated), 144, 480, 1152, 320
    ssd_taps = (("res2b", 75 * 75 * 32), ("res3a", 38 * 38 * 128), ("res5b", 19 * 19 * 256), ("conv7_2", 5 * 5 * 128),
                ("norm3.head", 38 * 38 * 24))
    h.set_option("fuse_expand", 0)          # the expand convs of blocks 1-5 run as GEMMs too (K = 16, 24, 40)
    base = None
    try:
        for i in range(-1, count):
            h.set_option("gemm_tile", i)
            got = {"logits": h.classify(x)}
            for t in taps:
                got[t] = h.tap(xd.ptr, 3, t, 3 * 112 * 112 * 96).copy()
            got["b2.exp"] = h.tap(xd.ptr, 3, "b2.exp", 3 * 56 * 56 * 144).copy()
            for t, cap in ssd_taps:
                got["ssd." + t] = h.ssd_tap(frame, t, cap).copy()
            got["boxes"] = np.asarray(h.detect_faces(frame, 0.3), np.int64)
            if base is None:
                base = got
                continue
            for k, v in got.items():
                assert v.shape == base[k].shape and np.array_equal(v, base[k]), f"tile {i}: {k} differs"
    finally:
        h.set_option("gemm_tile", -1)
        h.set_option("fuse_expand", 1)
        xd.free()
    want = b0_ref.forward(pkg.weights.to_torch(seeded_sd), torch.from_numpy(x)).numpy()
    assert np.abs(base["logits"] - want).max() <= 1e-3
