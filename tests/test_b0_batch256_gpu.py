"""GPU parity at the size bench.py runs (BASELINE.json configs[1], SURVEY section 8(d) Config 2):
torch.manual_seed(1); randn(256,3,224,224) on a max_batch=256 handle whose GEMM tiles were measured by dfd_warmup
at batch 256 - i.e. the kernels and tiles of the benchmark, not those of a batch-3 run."""
import numpy as np
import pytest
import torch

import frames as F
from oracle import b0_ref

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def big(pkg, seeded_sd, ssd_sd):
    blob = pkg.weights.pack_all(seeded_sd, ssd_sd)
    h = pkg._lib.Handle(blob, device=0, max_batch=256)
    h.warmup(256, 64)
    yield h
    h.close()


@pytest.fixture(scope="module")
def crops():
    torch.manual_seed(1)
    return torch.randn(256, 3, 224, 224)


def test_batch256_logits_match_oracle_and_chunks_of_8(pkg, big, crops, seeded_sd):
    x = crops.numpy()
    full = big.classify(x)
    assert full.shape == (256, 1) and np.all(np.isfinite(full))
    sd = pkg.weights.to_torch(seeded_sd)
    for lo in (0, 248):
        want = b0_ref.forward(sd, crops[lo:lo + 8]).numpy()
        err = float(np.abs(full[lo:lo + 8] - want).max())
        assert err <= LOGIT_TOL, (lo, err)
    assert np.ptp(full) > 0.05
    # bit-equality of all 256 logits with the same crops run 8 at a time (other M, other tiles, other grid sizes)
    parts = np.concatenate([big.classify(x[i:i + 8]) for i in range(0, 256, 8)])
    assert np.array_equal(parts, full)
    assert np.array_equal(big.classify(x), full)           # run-to-run


def test_batch256_every_tile_identical(pkg, big, crops):
    """the tile sweep of test_gemm_tiles_gpu at the benchmark's M (256*49 ... 256*12544 rows)"""
    x = crops.numpy()
    xd = big.alloc(x.nbytes).upload(x)
    yd = big.alloc(256 * 4)
    try:
        big.classify_device(xd.ptr, 256, yd.ptr)
        base = yd.download((256, 1))
        for i in range(pkg._lib.load().dfd_gemm_tile_count()):
            big.set_option("gemm_tile", i)
            big.classify_device(xd.ptr, 256, yd.ptr)
            assert np.array_equal(yd.download((256, 1)), base), f"tile {i}"
    finally:
        big.set_option("gemm_tile", -1)
        xd.free()
        yd.free()


def test_analyze_batch_1080p_x64_matches_per_frame_path(big):
    """dfd_analyze_batch_device at the e2e benchmark's size (64 x 1080p frames, 4 forced boxes per frame + the
    detector on every frame) against the per-frame entry points on a sample of frames: boxes identical, logits
    bit-equal (same kernels, other batch), forensic probability identical to the stateless per-frame value."""
    H, W, K, N = 1080, 1920, 4, 64
    rng = np.random.default_rng(7)
    frames = rng.integers(50, 200, (N, H, W, 3), dtype=np.uint8)
    for f in (0, 17, 40, 63):                      # give the sampled frames structure so the detector fires
        frames[f, 200:200 + 480, 300:300 + 640] = F.natural_like(480, 640, seed=30 + f)
    forced = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * N
    fd = big.alloc(frames.nbytes).upload(frames)
    try:
        fb, fl, fp = big.analyze_batch_device(fd.ptr, N, H, W, forced_boxes=forced, max_faces=K, with_forensics=True)
        db, dl, _ = big.analyze_batch_device(fd.ptr, N, H, W, forced_boxes=None, confidence_threshold=0.3, max_faces=K,
                                             with_forensics=False)
    finally:
        fd.free()
    assert all(len(b) == K for b in fb)
    for f in (0, 17, 40, 63):
        one = big.classify_crops(frames[f], forced[f]).reshape(-1)
        assert np.array_equal(one, fl[f]), (f, one, fl[f])
        det = big.detect_faces(frames[f], 0.3)
        assert db[f] == det[:K], (f, db[f], det[:K])
        if det:
            one = big.classify_crops(frames[f], det[:K]).reshape(-1)
            assert np.array_equal(one, dl[f])
        big.forensics_reset(900 + f)
        _, prob, _ = big.forensics(frames[f], True, stream_id=900 + f)
        assert prob == fp[f], (f, prob, fp[f])
    assert sum(len(b) for b in db) > 0, "the detector never fired: the detected-box half of this test is empty"


def test_batch256_late_block_launches_agree_with_the_default_path(big, crops):
    """Option "fuse_late" at the benchmark's size: blocks 6-10 / 12-15 as whole-image launches.  The expand products are
    the same six exact bf16 cross terms as in the GEMM, summed in a different order: logits agree to fp32 round-off."""
    x = crops.numpy()
    base = big.classify(x)
    big.set_option("fuse_late", 1)
    try:
        late = big.classify(x)
    finally:
        big.set_option("fuse_late", 0)
    assert np.all(np.isfinite(late))
    assert float(np.abs(late - base).max()) <= 1e-4
