"""CPU: detector oracle structure and the reference's guards (face_detection.py:51-68)."""
import numpy as np

import frames as F
from oracle import ssd_ref


def test_topology_counts(pkg):
    S = pkg.ssd_arch
    assert S.num_priors() == 8732
    sh = S.shapes()
    assert [sh[s[0]][1] for s in S.SOURCES] == [38, 19, 10, 5, 3, 1]
    assert [S.priors_per_cell(s[5]) for s in S.SOURCES] == [4, 6, 6, 6, 4, 4]
    pri = ssd_ref.prior_boxes(S.SOURCES, S.INPUT)
    assert pri.shape == (8732, 4)
    # first cell of the first map: 30x30 box centred at (4,4), then sqrt(30*60), then 2:1 and 1:2
    assert np.allclose(pri[0] * 300, [4 - 15, 4 - 15, 4 + 15, 4 + 15])
    assert np.allclose((pri[1, 2] - pri[1, 0]) * 300, np.sqrt(1800.0), atol=1e-4)
    assert np.allclose((pri[2, 2] - pri[2, 0]) / (pri[2, 3] - pri[2, 1]), 2.0, atol=1e-5)
    r = np.sqrt(2.0)                                               # last prior: min 264 at aspect 1:2
    assert np.allclose(pri[-1] * 300, [150 - 132 / r, 150 - 132 * r, 150 + 132 / r, 150 + 132 * r], atol=1e-3)


def test_guards_return_empty_lists(pkg, ssd_sd):
    sd = pkg.weights.to_torch(ssd_sd)
    S = pkg.ssd_arch
    assert ssd_ref.detect_bounding_box(sd, S, None) == []
    assert ssd_ref.detect_bounding_box(sd, S, np.zeros(100, np.uint8)) == []
    assert ssd_ref.detect_bounding_box(sd, S, np.zeros((10, 10, 3), np.uint8)) == []
    assert ssd_ref.detect_bounding_box(sd, S, np.zeros((0, 0, 3), np.uint8)) == []
    out = ssd_ref.detect_bounding_box(sd, S, F.face_frame())
    assert isinstance(out, list) and all(len(b) == 4 and b[2] > 20 and b[3] > 20 for b in out)


def test_postprocess_integer_rules():
    rows = [(0.9, 0.1, 0.1, 0.5, 0.5), (0.5, 0.0, 0.0, 1.0, 1.0), (0.8, -0.2, -0.1, 1.3, 1.2),
            (0.7, 0.5, 0.5, 0.51, 0.9), (0.95, 0.2999, 0.2, 0.7001, 0.8)]
    out = ssd_ref.postprocess(rows, 100, 200, 0.5)
    # row 1: conf == thr is rejected (strict >); row 2: clamped to the frame; row 3: 2 px wide -> dropped
    assert out == [(20, 10, 80, 40), (0, 0, 200, 100), (59, 20, 81, 60)]


def test_nms_keeps_best_of_overlapping_and_is_score_ordered():
    boxes = np.array([[0, 0, 1, 1], [0.05, 0, 1.05, 1], [2, 2, 3, 3], [0, 0, 1, 0.5]], np.float32)
    prob = np.array([0.6, 0.9, 0.7, 0.8], np.float32)
    rows = ssd_ref.detection_output(boxes, prob, 0.01, 0.45, 400, 200)
    assert [round(r[0], 1) for r in rows] == [0.9, 0.7]          # 0.8 (IoU 0.475 with the 0.9 box) and 0.6 are suppressed
    rows = ssd_ref.detection_output(boxes, prob, 0.01, 0.5, 400, 200)
    assert [round(r[0], 1) for r in rows] == [0.9, 0.8, 0.7]
