"""Committed golden vectors (tests/golden/*.json, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP path hits the same numbers through the C ABI."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import frames as F
from oracle import b0_ref, ssd_ref
from oracle.forensics_ref import ForensicsRef

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)

B0 = json.load(open(os.path.join(G, "b0_logits.json")))
FOR = json.load(open(os.path.join(G, "forensic_scores.json")))
SSD = json.load(open(os.path.join(G, "ssd_boxes.json")))
MT = json.load(open(os.path.join(G, "mtcnn_faces.json")))


def test_oracle_reproduces_b0_golden(pkg, seeded_sd):
    y = b0_ref.forward(pkg.weights.to_torch(seeded_sd), torch.from_numpy(mg.b0_inputs()))
    assert np.abs(y.numpy().ravel() - np.array(B0["logits"])).max() <= 1e-4


def test_oracle_reproduces_forensic_golden():
    for name in ("determinism", "gradient"):
        a = ForensicsRef()
        r = a.analyze(mg.FORENSIC_FRAMES[name]())
        assert r["scores"] == FOR[name]["scores"] and r["fake_probability"] == FOR[name]["fake_probability"]


def test_oracle_reproduces_ssd_golden(pkg, ssd_sd):
    f = F.natural_like()
    rows = ssd_ref.forward(pkg.weights.to_torch(ssd_sd), pkg.ssd_arch, f)
    assert [list(b) for b in ssd_ref.postprocess(rows, f.shape[0], f.shape[1], 0.5)] == SSD["frames"]["natural_720p"]["boxes"]


@pytest.mark.gpu
def test_hip_hits_b0_golden(b0_handle):
    got = b0_handle.classify(mg.b0_inputs())
    assert np.abs(got.ravel() - np.array(B0["logits"])).max() <= 1e-3
    torch.manual_seed(42)
    one = b0_handle.classify(torch.randn(1, 3, 224, 224).numpy())
    assert abs(float(one[0, 0]) - B0["reference_determinism_input_logit"]) <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FOR))
def test_hip_hits_forensic_golden(b0_handle, name):
    b0_handle.forensics_reset(950)
    scores, prob, stats = b0_handle.forensics(mg.FORENSIC_FRAMES[name](), True, 950)
    assert scores == FOR[name]["scores"] and prob == FOR[name]["fake_probability"]
    for k, v in FOR[name]["stats"].items():
        assert abs(stats[k] - v) <= 2e-4 * max(1.0, abs(v)), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SSD["frames"]))
def test_hip_hits_ssd_golden(b0_handle, name):
    got = b0_handle.detect_faces(mg.SSD_FRAMES[name](), 0.5)
    assert [list(b) for b in got] == SSD["frames"][name]["boxes"]


def _mt_case(i):
    from tests import mt_images

    return mt_images.images()[i], MT["cases"][i]


@pytest.mark.parametrize("i", range(len(MT["cases"])))
def test_oracle_reproduces_mtcnn_golden(pkg, mtcnn_sd, i):
    from oracle import mtcnn_ref

    img, want = _mt_case(i)
    taps = {}
    face = mtcnn_ref.mtcnn_forward(pkg.weights.to_torch(mtcnn_sd), img, taps)
    assert [len(taps[k]) for k in ("stage1", "stage2", "stage3")] == want["rows"]
    assert (face is None) == (want["selected"] is None)
    if face is not None:
        assert np.abs(np.array(want["selected"]) - taps["selected"]).max() <= 1e-2
        assert int(face.sum()) == want["face_sum"] and [int(v) for v in face[:, 0, :4].ravel()] == want["face_corner"]


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(MT["cases"])))
def test_hip_hits_mtcnn_golden(mt_handle, i):
    img, want = _mt_case(i)
    bgr = np.ascontiguousarray(img[..., ::-1])
    assert [mt_handle.mtcnn_tap(bgr, k).shape[0] for k in ("stage1", "stage2", "stage3")] == want["rows"]
    face, box = mt_handle.mtcnn_align(bgr)
    assert (face is None) == (want["selected"] is None)
    if face is not None:
        assert np.abs(np.array(want["selected"]) - box).max() <= 1e-2
        assert int(face.sum()) == want["face_sum"] and [int(v) for v in face[:, 0, :4].ravel()] == want["face_corner"]
