"""GPU parity for the SSD-style detector through the C ABI vs oracle/ssd_ref.py: every trunk
tensor and head, per-prior probabilities and decoded boxes, the DetectionOutput rows, and the
reference's integer boxes (bit-exact except where the oracle's float coordinate lies within
1e-3 px of an integer, which the test reports instead of hiding)."""
import numpy as np
import pytest
import torch

import frames as F
from oracle import ssd_ref

pytestmark = pytest.mark.gpu

ACT_TOL = 2e-4          # activations are O(1); fp32 MFMA vs torch-CPU summation order


@pytest.fixture(scope="module")
def tsd(pkg, ssd_sd):
    return pkg.weights.to_torch(ssd_sd)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def test_every_tensor_matches_oracle(pkg, b0_handle, tsd):
    S = pkg.ssd_arch
    frame = F.face_frame(1280, 720, 3)
    taps = {}
    ssd_ref.forward(tsd, S, frame, taps)
    for name, _, _ in S.LAYERS:
        want = _nhwc(taps[name])
        got = b0_handle.ssd_tap(frame, name, want.size).reshape(want.shape)
        err = np.abs(got - want).max()
        assert err <= ACT_TOL * max(1.0, np.abs(want).max()), (name, err)
    off = 0
    for src, c, m, _, _, ars, _ in S.SOURCES:
        p = S.priors_per_cell(ars)
        got = b0_handle.ssd_tap(frame, src + ".head", m * m * p * 6).reshape(m * m, p * 6)
        n = m * m * p
        assert np.abs(got[:, : p * 4].reshape(-1, 4) - taps["loc"][off:off + n]).max() <= ACT_TOL * 4
        assert np.abs(got[:, p * 4:].reshape(-1, 2) - taps["conf"][off:off + n]).max() <= ACT_TOL * 8
        off += n
    prob = b0_handle.ssd_tap(frame, "prob", 8732)
    boxes = b0_handle.ssd_tap(frame, "boxes", 8732 * 4).reshape(8732, 4)
    assert np.abs(prob - taps["prob"]).max() <= 1e-4
    assert np.abs(boxes - taps["boxes"]).max() <= 1e-4


@pytest.mark.parametrize("name,frame", [("face_vga", F.face_frame()), ("face_1080p", F.face_frame(1920, 1080, 5)),
                                        ("natural", F.natural_like()), ("blank", F.blank_frame()),
                                        ("noise_qvga", F.noisy_image((240, 320), 9))])
def test_rows_and_boxes_match_oracle(pkg, b0_handle, tsd, name, frame):
    S = pkg.ssd_arch
    rows_ref = ssd_ref.forward(tsd, S, frame)
    rows = b0_handle.ssd_tap(frame, "rows", 200 * 5).reshape(-1, 5)
    assert len(rows) == len(rows_ref), (len(rows), len(rows_ref))
    if len(rows_ref):
        assert np.abs(rows - np.asarray(rows_ref, np.float32)).max() <= 2e-4
        assert np.all(np.diff(rows[:, 0]) <= 0)                          # descending confidence
    h, w = frame.shape[:2]
    want = ssd_ref.postprocess(rows_ref, h, w, 0.5)
    got = b0_handle.detect_faces(frame, 0.5)
    # coordinates whose float value sits on an integer boundary may legitimately truncate differently
    edge = 0
    for r in rows_ref:
        if r[0] > 0.5:
            for v, s in zip(r[1:], (w, h, w, h)):
                edge += abs(v * s - round(v * s)) < 1e-3 * max(1.0, s * 2e-4 / 1e-3)
    if edge == 0:
        assert got == want, (got[:5], want[:5])
    else:
        assert len(got) == len(want) and np.abs(np.asarray(got) - np.asarray(want)).max() <= 1
    assert all(bw > 20 and bh > 20 and x >= 0 and y >= 0 and x + bw <= w and y + bh <= h for (x, y, bw, bh) in got)


def test_guards_and_thresholds(b0_handle):
    assert b0_handle.detect_faces(np.zeros((10, 10, 3), np.uint8)) == []           # reference :55-56
    assert b0_handle.detect_faces(np.zeros((29, 400, 3), np.uint8)) == []
    f = F.face_frame()
    lo, hi = b0_handle.detect_faces(f, 0.3), b0_handle.detect_faces(f, 0.6)
    assert len(lo) >= len(hi)
    boxes, conf = b0_handle.detect_faces(f, 0.5, with_conf=True)
    assert len(boxes) == len(conf) and np.all(conf > 0.5) and np.all(np.diff(conf) <= 0)
    assert b0_handle.detect_faces(f, 0.5) == b0_handle.detect_faces(f, 0.5)         # deterministic
    assert b0_handle.detect_faces(f, 0.5, max_out=3) == boxes[:3]


def test_detection_output_wait_expiry_is_loud():
    """ssd_nms_kernel's walker waits (bounded) for the overlap rows its producer waves publish.  If that bound ever
    expired the walk used to go on over unpublished rows and return wrong boxes with rc 0 (ADVICE
    r3).  DFD_NMS_SPIN_BOUND=0 makes every wait "expire at once": the image's count comes back as -1 and the detector
    entry points must raise DFD_ERR_HIP (-3) - for one frame, for a batch, and for the rows tap - instead of returning boxes.
    Own process: the bound is read once per process."""
    import os
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import rtdfd_amd as pkg, frames as F
W = pkg.weights
h = pkg._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0)), device=0, max_batch=8)
f = F.face_frame()
codes = []
for fn in (lambda: h.detect_faces(f, 0.3), lambda: h.ssd_tap(f, "rows", 200 * 5),
           lambda: h.analyze_batch_device(h.alloc(f.nbytes).upload(f).ptr, 1, f.shape[0], f.shape[1], max_faces=4)):
    try:
        fn()
        codes.append(0)
    except pkg._lib.DfdError as e:
        codes.append(e.code)
        assert "overlap rows" in str(e), str(e)
print("codes", codes)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DFD_NMS_SPIN_BOUND="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "codes [-3, -3, -3]" in r.stdout, r.stdout
