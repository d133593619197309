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
    """Option "fuse_late" at the benchmark's size: blocks 6-10 / 12-15 as whole-image launches (the default since round 4)
    against expand GEMM + depthwise kernel as separate launches (round 3's default).  The expand products are the same
    six exact bf16 cross terms as in the GEMM, summed in a different order: logits agree to fp32 round-off."""
    x = crops.numpy()
    late = big.classify(x)
    big.set_option("fuse_late", 0)
    try:
        base = big.classify(x)
    finally:
        big.set_option("fuse_late", 1)
    assert np.all(np.isfinite(late))
    assert float(np.abs(late - base).max()) <= 1e-4


def test_config4_one_workload_1080p_x64_forensics_bf16(pkg, big):
    """BASELINE.json configs[3] / SURVEY 8(d) Config 4 as ONE workload: dfd_analyze_batch_device at
    64 x 1080p frames with the six forensic signals AND bf16 activation storage, against the fp32 run of the same call.
    Boxes and forensic probabilities identical (detector, CLAHE and the forensic kernels stay integer / fp32), NaN
    positions equal, logits within the bf16 statistical bars of tests/test_b0_bf16_gpu.py (rms / p95 / max over the 256
    crops), and the vote table bench.py prints (`bf16.vote_gate`): flipped votes at the reference's thresholds 0.5 / 0.55
    and at 32 quantile thresholds - zero wherever no fp32 probability lies within the largest bf16 error of the threshold."""
    import bench

    H, W, K, N = 1080, 1920, 4, 64
    rng = np.random.default_rng(7)
    frames = rng.integers(50, 200, (N, H, W, 3), dtype=np.uint8)
    for f in range(0, N, 5):                       # structure so that the detector fires on some frames
        frames[f, 200:200 + 480, 300:300 + 640] = F.natural_like(480, 640, seed=60 + f)
    forced = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * N
    fd = big.alloc(frames.nbytes).upload(frames)
    try:
        b32, l32, p32 = big.analyze_batch_device(fd.ptr, N, H, W, forced_boxes=forced, max_faces=K, with_forensics=True)
        d32, dl32, _ = big.analyze_batch_device(fd.ptr, N, H, W, confidence_threshold=0.3, max_faces=K, with_forensics=True)
        big.set_option("bf16_activations", 1)
        big.warmup(256, 0)
        b16, l16, p16 = big.analyze_batch_device(fd.ptr, N, H, W, forced_boxes=forced, max_faces=K, with_forensics=True)
        d16, dl16, q16 = big.analyze_batch_device(fd.ptr, N, H, W, confidence_threshold=0.3, max_faces=K, with_forensics=True)
    finally:
        big.set_option("bf16_activations", 0)
        fd.free()
    assert b16 == b32 and d16 == d32                                   # boxes: bit-exact, bf16 never touches the detector
    assert sum(len(b) for b in d32) > 0
    assert np.array_equal(np.asarray(p16), np.asarray(p32)) and np.array_equal(np.asarray(q16), np.asarray(p32))
    a = np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in l32])
    b = np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in l16])
    assert a.size == N * K and np.array_equal(np.isnan(a), np.isnan(b)) and not np.isnan(a).any()
    e = np.abs(a - b)
    rms, p95, mx = float(np.sqrt(np.mean(e ** 2))), float(np.quantile(e, 0.95)), float(e.max())
    print(f"configs[3] joint workload: bf16 vs fp32 logits over {a.size} crops: rms {rms:.2e} p95 {p95:.2e} max {mx:.2e}")
    assert rms <= 1.6e-2 and p95 <= 3.5e-2 and mx <= 1e-1 and mx > 1e-6
    for x, y in zip(dl32, dl16):                                       # detected boxes: same bars per crop
        if len(x):
            assert float(np.abs(np.asarray(x) - np.asarray(y)).max()) <= 1e-1
    gate = bench.vote_gate(a, b, frames=a.size)
    print("threshold flipped_votes flipped_verdict_frames frames_within_bf16_error")
    for r in gate["per_threshold"]:
        print(f"  {r['threshold']:.5f} {r['flipped_votes']:3d} {r['flipped_verdict_frames']:3d} {r['frames_within_bf16_error']:3d}")
        if r["frames_within_bf16_error"] == 0:
            assert r["flipped_votes"] == 0 and r["flipped_verdict_frames"] == 0, r
        assert r["flipped_votes"] <= r["frames_within_bf16_error"], r
    assert gate["per_threshold"][0]["threshold"] == 0.5 and gate["per_threshold"][1]["threshold"] == 0.55


def test_two_forwards_in_flight_return_the_single_forward_bits(pkg, big, crops, seeded_sd, ssd_sd):
    """bench.py's headline loop keeps two batch-256 forwards in flight (`ClassifierLanes`: a second handle with its own
    stream and workspace).  Different inputs per lane, interleaved without a wait between them: every lane's logits equal
    the single-handle run of the same crops bit for bit (other workspace, concurrent kernels, shared tile table)."""
    x = crops.numpy()
    xs = [x, np.ascontiguousarray(x[::-1]), x * np.float32(0.5)]
    want = [big.classify(a) for a in xs]
    lanes = pkg._lib.ClassifierLanes(pkg.weights.pack_all(seeded_sd, ssd_sd), device=0, max_batch=256, lanes=2, first=big)
    bufs = []
    try:
        lanes.warmup(256)
        assert lanes.handles[1].tiles_export() == big.tiles_export()
        xd = [big.alloc(a.nbytes).upload(a) for a in xs]
        yd = [big.alloc(256 * 4) for _ in range(6)]
        bufs = xd + yd
        # six forwards, no host wait in between; the third one ordered alone on the device (dfd_wait_for both ways: what
        # bench.py does with the step that carries its per-launch events)
        went = [(lanes.submit_alone if i == 2 else lanes.submit)(xd[i % 3].ptr, 256, yd[i].ptr) for i in range(6)]
        lanes.sync()
        assert went == [0, 1, 0, 1, 0, 1]
        for i in range(6):
            assert np.array_equal(yd[i].download((256, 1)), want[i % 3]), i
    finally:
        for b in bufs:
            b.free()
        lanes.close()


def test_wait_for_orders_two_handles_on_the_device(pkg, big, crops, seeded_sd, ssd_sd):
    """dfd_wait_for: handle B's work starts after what handle A has queued.  A writes logits into a buffer, B is told to
    wait for A and then classifies INTO THE SAME buffer from other crops: the buffer ends with B's result every time; the
    reverse order ends with A's.  (Without the ordering the two forwards run side by side and either could finish last.)"""
    x = crops.numpy()
    xa, xb = x, x * np.float32(0.25)
    wa, wb = big.classify(xa), big.classify(xb)
    assert not np.array_equal(wa, wb)
    other = pkg._lib.Handle(pkg.weights.pack_all(seeded_sd, ssd_sd), device=0, max_batch=256)
    bufs = []
    try:
        other.tiles_import(big.tiles_export())
        other.warmup(256, 0)
        # the main stream re-created in the high-priority queue pool (what a second worker's handle is given so that it
        # cannot share a hardware queue with the first): results and ordering unchanged; other values are rejected
        other.set_option("stream_priority", 1)
        with pytest.raises(pkg._lib.DfdError):
            other.set_option("stream_priority", 2)
        assert np.array_equal(other.classify(xb), wb)
        da, db, y = big.alloc(xa.nbytes).upload(xa), big.alloc(xb.nbytes).upload(xb), big.alloc(256 * 4)
        bufs = [da, db, y]
        for _ in range(3):
            big.classify_device(da.ptr, 256, y.ptr)
            other.wait_for(big)
            other.classify_device(db.ptr, 256, y.ptr)
            other.sync()
            assert np.array_equal(y.download((256, 1)), wb)
            big.wait_for(other)
            big.classify_device(da.ptr, 256, y.ptr)
            big.sync()
            assert np.array_equal(y.download((256, 1)), wa)
        big.wait_for(big)                                         # a handle and itself: no-op
    finally:
        big.sync(); other.sync()
        for b in bufs:
            b.free()
        other.close()


def test_lanes_overlap_check_keeps_the_faster_setting_and_the_bits(pkg, big, crops, seeded_sd, ssd_sd):
    """`ClassifierLanes.check_overlap` (bench.py runs it untimed): the normal path reports both timings and leaves lane 1
    alone; with a threshold no measurement can meet (gain 0) it tries the high-priority pool and keeps whichever was faster.
    Either way the lanes' logits stay the single-handle bits."""
    x = crops.numpy()
    want = big.classify(x)
    lanes = pkg._lib.ClassifierLanes(pkg.weights.pack_all(seeded_sd, ssd_sd), device=0, max_batch=256, lanes=2, first=big)
    bufs = []
    try:
        lanes.warmup(256)
        xd = big.alloc(x.nbytes).upload(x)
        ys = [big.alloc(256 * 4) for _ in range(2)]
        bufs = [xd] + ys
        rep = lanes.check_overlap(xd.ptr, 256, ys, steps=4)
        assert rep["one_lane_ms"] > 0 and rep["lanes_ms"] > 0 and rep["lane1_priority"] in (0, 1)
        forced = lanes.check_overlap(xd.ptr, 256, ys, steps=4, gain=0.0)
        assert "lanes_ms_high_priority" in forced and forced["lane1_priority"] in (0, 1)
        assert (forced["lane1_priority"] == 1) == (forced["lanes_ms_high_priority"] < forced["lanes_ms"])
        for i in range(4):
            lanes.submit(xd.ptr, 256, ys[i % 2].ptr)
        lanes.sync()
        for y in ys:
            assert np.array_equal(y.download((256, 1)), want)
    finally:
        lanes.handles[1].set_option("stream_priority", 0)
        for b in bufs:
            b.free()
        lanes.close()
