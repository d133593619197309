"""GPU: the batched device-resident path equals the per-frame path frame by frame (same boxes,
same logits bit for bit, same stateless forensic probability)."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu


def test_batch_equals_per_frame(b0_handle):
    frames = [F.natural_like(480, 640, 31), F.blank_frame(640, 480), F.natural_like(480, 640, 32),
              F.face_frame(640, 480, 3), F.natural_like(480, 640, 33)]
    n = len(frames)
    stack = np.ascontiguousarray(np.stack(frames))
    dev = b0_handle.alloc(stack.nbytes).upload(stack)
    boxes, logits, fprob = b0_handle.analyze_batch_device(dev.ptr, n, 480, 640, max_faces=6, with_forensics=True)
    for f, frame in enumerate(frames):
        want_boxes = b0_handle.detect_faces(frame, 0.5)[:6]
        assert boxes[f] == want_boxes
        if want_boxes:
            want_logits = b0_handle.classify_crops(frame, want_boxes, apply_clahe=True).ravel()
            assert np.array_equal(logits[f], want_logits)
        b0_handle.forensics_reset(960)
        _, p, _ = b0_handle.forensics(frame, True, 960)               # first frame of a fresh stream: temporal = 0
        assert fprob[f] == p
    # forced boxes: the detector still runs, the crops are the caller's
    forced = [[(10, 20, 230, 240), (300, 100, 224, 224)]] * n
    fb, fl, _ = b0_handle.analyze_batch_device(dev.ptr, n, 480, 640, forced_boxes=forced, max_faces=2)
    for f, frame in enumerate(frames):
        assert fb[f] == forced[f]
        assert np.array_equal(fl[f], b0_handle.classify_crops(frame, forced[f], apply_clahe=True).ravel())
    dev.free()


def test_batch_larger_than_classifier_capacity(b0_handle):
    """more crops than max_batch (16) are classified in chunks"""
    frames = np.ascontiguousarray(np.stack([F.natural_like(300, 400, 40 + i) for i in range(6)]))
    dev = b0_handle.alloc(frames.nbytes).upload(frames)
    forced = [[(0, 0, 100, 100), (50, 50, 224, 224), (100, 20, 250, 260), (10, 10, 64, 64)]] * 6       # 24 crops
    fb, fl, _ = b0_handle.analyze_batch_device(dev.ptr, 6, 300, 400, forced_boxes=forced, max_faces=4)
    for f in range(6):
        assert np.array_equal(fl[f], b0_handle.classify_crops(frames[f], forced[f], apply_clahe=True).ravel())
    dev.free()


@pytest.mark.parametrize("h,w", [(353, 517), (31, 45), (721, 1283)])
def test_batch_ragged_frame_sizes(b0_handle, h, w):
    """odd frame geometry (no dimension a multiple of anything, one just above the detector's 30-pixel guard) and crops
    whose byte size is not a multiple of 4 (the CLAHE kernel's vector path ends in a short group): batch == per-frame"""
    frames = np.ascontiguousarray(np.stack([F.natural_like(h, w, 70 + i) for i in range(3)]))
    dev = b0_handle.alloc(frames.nbytes).upload(frames)
    forced = [[(1, 2, min(w - 2, 37), min(h - 3, 29)), (w // 3, h // 4, min(w - w // 3, 101), min(h - h // 4, 83))]] * 3
    boxes, logits, fprob = b0_handle.analyze_batch_device(dev.ptr, 3, h, w, forced_boxes=forced, max_faces=2, with_forensics=True)
    for f in range(3):
        assert boxes[f] == forced[f]
        assert np.array_equal(logits[f], b0_handle.classify_crops(frames[f], forced[f], apply_clahe=True).ravel())
        b0_handle.forensics_reset(961)
        _, p, _ = b0_handle.forensics(frames[f], True, 961)
        assert fprob[f] == p
    free = b0_handle.analyze_batch_device(dev.ptr, 3, h, w, max_faces=3)[0]
    for f in range(3):
        assert free[f] == b0_handle.detect_faces(frames[f], 0.5)[:3]
    dev.free()


def test_forensics_beside_the_detector_equals_forensics_in_front(b0_handle):
    """the batch call runs the six-signal launch set on a second stream beside detector / classifier (default) or in front
    of them on the main stream (option overlap_forensics = 0): the same probabilities, boxes and logits, call after call
    (the second stream's buffers and events are reused)"""
    frames = np.ascontiguousarray(np.stack([F.natural_like(480, 640, 90 + i) for i in range(7)]))
    dev = b0_handle.alloc(frames.nbytes).upload(frames)
    got = {}
    try:
        for flag in (1, 0, 1):
            b0_handle.set_option("overlap_forensics", flag)
            got.setdefault(flag, []).append(b0_handle.analyze_batch_device(dev.ptr, 7, 480, 640, max_faces=4, with_forensics=True))
            got[flag].append(b0_handle.analyze_batch_device(dev.ptr, 5, 480, 640, max_faces=4, with_forensics=True))
    finally:
        b0_handle.set_option("overlap_forensics", 1)
    ref = got[0]
    for k, run in enumerate(got[1]):
        boxes, logits, fprob = run
        rb, rl, rp = ref[k % 2]
        assert boxes == rb and list(fprob) == list(rp)
        for a, b in zip(logits, rl):
            assert np.array_equal(a, b)
    dev.free()
