"""GPU parity of the MTCNN align/crop stage through the C ABI vs oracle/mtcnn_ref.py.

Bars: network maps |d| <= 1e-4 (fp32, different summation order than torch's conv); stage boxes the same
rows to 1e-2 px (float32 box arithmetic on 1e-4-accurate regressions); the selected 160x160 crop bit-exact
(integer box corners, Pillow-exact 8-bit resize); the logit of the aligned crop within the classifier's 1e-3.
A cell or candidate whose probability is within 1e-4 of its threshold may legitimately fall either way:
the test asserts that no cell does with the seeded weights/images and says so if one ever does."""
import numpy as np
import pytest
import torch

from oracle import b0_ref, imgproc_ref, mtcnn_ref as M
from tests import mt_images

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd(pkg, mtcnn_sd):
    return pkg.weights.to_torch(mtcnn_sd)


def _bgr(rgb):
    return np.ascontiguousarray(rgb[..., ::-1])


@pytest.mark.parametrize("idx", range(len(mt_images.CASES)))
def test_cascade_matches_oracle(mt_handle, sd, idx):
    rgb = mt_images.images()[idx]
    taps = {}
    want = M.mtcnn_forward(sd, rgb, taps)
    bgr = _bgr(rgb)
    assert mt_handle.has_mtcnn
    for si in range(len(M.scale_pyramid(*rgb.shape[:2]))):
        p = mt_handle.mtcnn_tap(bgr, f"pnet.prob.{si}")
        r = mt_handle.mtcnn_tap(bgr, f"pnet.reg.{si}")
        wp, wr = taps[f"pnet.prob.{si}"], taps[f"pnet.reg.{si}"].transpose(1, 2, 0)
        assert p.shape == wp.shape and np.abs(p - wp).max() <= 1e-4, (si, p.shape, wp.shape)
        assert np.abs(r.reshape(wr.shape) - wr).max() <= 1e-4
        flip = (p >= np.float32(0.6)) != (wp >= np.float32(0.6))
        assert not flip.any(), (f"level {si}: {int(flip.sum())} cells fall on different sides of the P-Net threshold "
                                f"(margins {np.abs(wp[flip] - 0.6)}): threshold-ambiguous input, pick another seed")
    for stage in ("stage1", "stage2", "stage3"):
        got = mt_handle.mtcnn_tap(bgr, stage)
        w = taps[stage]
        assert got.shape[0] == len(w), (stage, got.shape, len(w))
        if len(w):
            assert np.abs(got[:, :4] - w[:, :4]).max() <= 1e-2 and np.abs(got[:, 4] - w[:, 4]).max() <= 1e-4, stage
    face, box = mt_handle.mtcnn_align(bgr)
    if want is None:
        assert face is None
    else:
        assert face is not None
        assert np.abs(box - taps["selected"]).max() <= 1e-2
        assert np.array_equal(face, want)               # bit-exact 160x160 crop, RGB planes 0..255


def test_undersized_and_option(mt_handle, sd):
    tiny = _bgr(mt_images.textured(12, 40, 8))
    assert mt_handle.mtcnn_align(tiny) == (None, None)


def test_classify_crops_with_alignment(pkg, mt_handle, sd, seeded_sd):
    """Integrated path (reference deepfake_detection.py:517-538 -> 372-398): CLAHE'd crop -> MTCNN.forward ->
    224 bilinear + normalise -> B0; NaN where the cascade finds no face; option "mtcnn"=0 restores the bypass."""
    rs = np.random.RandomState(11)
    frame = rs.randint(40, 215, (480, 640, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    frame = np.clip(frame * 0.4 + (110 + 60 * np.sin(xx / 19.0) * np.cos(yy / 27.0))[..., None] * 0.6, 0, 255).astype(np.uint8)
    boxes = np.array([[20, 30, 300, 280], [330, 40, 200, 180], [100, 330, 90, 75], [400, 300, 230, 170]], np.int32)
    got = mt_handle.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
    b0 = pkg.weights.to_torch(seeded_sd)
    want = []
    for x, y, w, h in boxes:
        crop = imgproc_ref.preprocess_face_quality(frame[y:y + h, x:x + w])
        face = M.mtcnn_forward(sd, np.ascontiguousarray(crop[..., ::-1]))
        if face is None:
            want.append(np.nan)
            continue
        face_bgr = np.ascontiguousarray(face.transpose(1, 2, 0)[..., ::-1]).astype(np.uint8)
        x224 = imgproc_ref.crop_resize_normalize(face_bgr)
        want.append(float(b0_ref.forward(b0, torch.from_numpy(x224[None])).reshape(-1)[0]))
    want = np.asarray(want, np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(want)), (got, want)
    ok = ~np.isnan(want)
    assert ok.any() and np.abs(got[ok] - want[ok]).max() <= 1e-3, (got, want)
    try:
        mt_handle.set_option("mtcnn", 0)
        plain = mt_handle.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
    finally:
        mt_handle.set_option("mtcnn", 1)
    assert not np.isnan(plain).any() and np.abs(plain[ok] - got[ok]).max() > 1e-3     # a different crop reaches B0


def test_selective_cascade_batch(pkg, seeded_sd):
    """The funnel of a trained cascade (weights.MTCNN_SELECTIVE: most pyramid cells and windows rejected) through the
    batched crop path: which crops keep a face, and their logits, equal the oracle's per-crop run.  A window whose
    R-/O-Net probability lies within 1e-4 of 0.7 may fall either way (the wide dense layers run on the split-precision
    GEMM, another summation order than the oracle's): such crops are reported and skipped, not hidden."""
    W = pkg.weights
    sel = W.seeded_mtcnn_state_dict(0, W.MTCNN_SELECTIVE)
    tsel = W.to_torch(sel)
    h = pkg._lib.Handle(W.pack_all(seeded_sd, W.seeded_ssd_state_dict(0), sel), device=0, max_batch=16)
    try:
        frame = np.random.default_rng(7).integers(50, 200, (2, 1080, 1920, 3), dtype=np.uint8)[0]     # the bench's frame 0
        boxes = np.array([[200, 150, 320, 400], [900, 300, 256, 256], [1400, 500, 400, 480], [600, 700, 224, 224],
                          [100, 600, 300, 300], [1200, 80, 280, 330]], np.int32)
        before = h.classifier_crop_count()
        got = h.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
        assert h.classifier_crop_count() - before == int((~np.isnan(got)).sum())      # rejected crops are never classified
        b0 = W.to_torch(seeded_sd)
        found_want, ambiguous = [], []
        for k, (x, y, w, hh) in enumerate(boxes):
            crop = imgproc_ref.preprocess_face_quality(frame[y:y + hh, x:x + w])
            taps = {}
            face = M.mtcnn_forward(tsel, np.ascontiguousarray(crop[..., ::-1]), taps)
            probs = np.concatenate([np.asarray(taps.get(n, []), np.float32).reshape(-1) for n in ("rnet.prob", "onet.prob")])
            if probs.size and np.abs(probs - 0.7).min() <= 1e-4:
                ambiguous.append(k)
            found_want.append(face is not None)
            if face is not None and k not in ambiguous:
                face_bgr = np.ascontiguousarray(face.transpose(1, 2, 0)[..., ::-1]).astype(np.uint8)
                want = float(b0_ref.forward(b0, torch.from_numpy(imgproc_ref.crop_resize_normalize(face_bgr)[None])).reshape(-1)[0])
                assert abs(got[k] - want) <= 1e-3, (k, got[k], want)
        keep = [k for k in range(len(boxes)) if k not in ambiguous]
        assert len(keep) >= 4, f"threshold-ambiguous crops {ambiguous}: pick another seed"
        assert [bool(~np.isnan(got[k])) for k in keep] == [found_want[k] for k in keep], (got, found_want, ambiguous)
        assert any(found_want) and not all(found_want)          # the funnel both passes and rejects crops
    finally:
        h.close()


def test_threaded_box_logic_equals_serial(pkg, mt_handle, monkeypatch):
    """A dense funnel (the random-init cascade on large crops: thousands of candidates per crop) sends the per-level and
    per-crop box logic to host threads; DFD_HOST_THREADS=1 keeps it on the calling thread.  Same logits, same crops
    without a face."""
    frame = np.random.default_rng(7).integers(50, 200, (2, 1080, 1920, 3), dtype=np.uint8)[0]
    boxes = np.array([[200, 150, 320, 400], [900, 300, 256, 256], [1400, 500, 400, 480], [600, 700, 224, 224]], np.int32)
    threaded = mt_handle.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
    monkeypatch.setenv("DFD_HOST_THREADS", "1")
    serial = mt_handle.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
    assert np.array_equal(threaded, serial, equal_nan=True), (threaded, serial)
    assert not np.isnan(threaded).all()


def test_device_box_logic_equals_host_path(pkg, mt_handle, monkeypatch):
    """The box bookkeeping between the networks runs on the device (mtcnn_boxes.hip: one block per crop and stage);
    DFD_MT_DEVICE_BOXES=0 keeps it on the library's host side.  Same arithmetic in the same order: every stage's rows,
    the selected box, the 160x160 crop and the logits are IDENTICAL, on the dense funnel (hundreds to thousands of
    candidates per crop, ties included) and on a crop without any level."""
    rs = np.random.RandomState(5)
    for hh, ww in ((150, 170), (96, 210), (230, 190), (12, 40)):
        bgr = _bgr(mt_images.textured(hh, ww, int(rs.randint(1 << 30))))
        got = {}
        for flag in ("1", "0"):
            monkeypatch.setenv("DFD_MT_DEVICE_BOXES", flag)
            rows = []
            if min(hh, ww) >= 20:
                for st, net in (("stage1", "rnet"), ("stage2", "onet"), ("stage3", None)):
                    rows.append(mt_handle.mtcnn_tap(bgr, st))
                    if net and len(rows[-1]):
                        rows += [mt_handle.mtcnn_tap(bgr, net + ".prob"), mt_handle.mtcnn_tap(bgr, net + ".reg")]
            got[flag] = (rows, mt_handle.mtcnn_align(bgr))
        (rd, (fd, bd)), (rh, (fh, bh)) = got["1"], got["0"]
        assert len(rd) == len(rh)
        for a, b in zip(rd, rh):
            assert a.shape == b.shape and np.array_equal(a, b), (hh, ww, a.shape, b.shape)
        if min(hh, ww) >= 20:
            assert len(rd[0]) > 20, "the dense cascade should hand many boxes to R-Net"
        assert (fd is None) == (fh is None)
        if fd is not None:
            assert np.array_equal(bd, bh) and np.array_equal(fd, fh)
    frame = np.random.default_rng(7).integers(50, 200, (2, 1080, 1920, 3), dtype=np.uint8)[0]
    boxes = np.array([[200, 150, 120, 140], [900, 300, 156, 156], [1400, 500, 100, 180], [600, 700, 224, 124]], np.int32)
    monkeypatch.setenv("DFD_MT_DEVICE_BOXES", "1")
    dev = mt_handle.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
    monkeypatch.setenv("DFD_MT_DEVICE_BOXES", "0")
    host = mt_handle.classify_crops(frame, boxes, apply_clahe=True).reshape(-1)
    assert np.array_equal(dev, host, equal_nan=True), (dev, host)
    assert not np.isnan(dev).all()


def test_box_capacity_overflow_falls_back_to_the_host_path(pkg, mt_handle, monkeypatch):
    """A crop with more P-Net candidates than a device block holds (the random-init cascade on a 700 x 900 crop: ~20 %
    of ~100k cells) raises the overflow flag; the step then runs on the host path - same result as with the device
    path switched off."""
    bgr = _bgr(mt_images.textured(700, 900, 3))
    monkeypatch.setenv("DFD_MT_DEVICE_BOXES", "1")
    fd, bd = mt_handle.mtcnn_align(bgr)
    monkeypatch.setenv("DFD_MT_DEVICE_BOXES", "0")
    fh, bh = mt_handle.mtcnn_align(bgr)
    assert (fd is None) == (fh is None)
    if fd is not None:
        assert np.array_equal(bd, bh) and np.array_equal(fd, fh)


def test_device_box_logic_at_the_bench_size(pkg, seeded_sd):
    """bench.py's MTCNN rows (64 x 1080p frames, 4 forced crops each, selective cascade): the 256 logits of the device box
    path equal the host path's bit for bit, NaN positions included - the property that holds at any size."""
    import os

    W = pkg.weights
    sel = W.seeded_mtcnn_state_dict(0, W.MTCNN_SELECTIVE)
    n = 64
    h = pkg._lib.Handle(W.pack_all(seeded_sd, W.seeded_ssd_state_dict(0), sel), device=0, max_batch=4 * n)
    try:
        frames = np.random.default_rng(7).integers(50, 200, (n, 1080, 1920, 3), dtype=np.uint8)
        boxes = [[(200, 150, 320, 400), (900, 300, 256, 256), (1400, 500, 400, 480), (600, 700, 224, 224)]] * n
        fd = h.alloc(frames.nbytes).upload(frames)
        got, classified = {}, {}
        for flag in ("1", "0"):
            os.environ["DFD_MT_DEVICE_BOXES"] = flag
            before = h.classifier_crop_count()
            res = h.analyze_batch_device(fd.ptr, n, 1080, 1920, forced_boxes=boxes, max_faces=4)
            classified[flag] = h.classifier_crop_count() - before
            got[flag] = np.concatenate([np.asarray(l, np.float32).reshape(-1) for l in res[1]])
        os.environ.pop("DFD_MT_DEVICE_BOXES", None)
        assert got["1"].size == 4 * n
        assert np.array_equal(got["1"], got["0"], equal_nan=True)
        found = int((~np.isnan(got["1"])).sum())
        assert 0 < found < 4 * n
        # reference deepfake_detection.py:377-380: `mtcnn()` -> None returns BEFORE the model runs.  A call with 4 n boxes
        # of which `found` keep a face runs the classifier at batch `found`, not 4 n (round 3 classified the zero-filled
        # faces of the rejected crops too and overwrote their logits with NaN afterwards)
        assert classified == {"1": found, "0": found}, (classified, found)
        h.set_option("mtcnn", 0)
        before = h.classifier_crop_count()
        res = h.analyze_batch_device(fd.ptr, n, 1080, 1920, forced_boxes=boxes, max_faces=4)
        assert h.classifier_crop_count() - before == 4 * n            # stage off: every box is classified
        h.set_option("mtcnn", 1)
        fd.free()
    finally:
        os.environ.pop("DFD_MT_DEVICE_BOXES", None)
        h.close()
