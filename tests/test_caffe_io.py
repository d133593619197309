"""a Caffe SSD face detector (deploy.prototxt + caffemodel, reference face_detection.py:19-24) read
without cv2 / protobuf and turned into the detector plan.  CPU: the parsers round-trip what tests/caffe_net.py
writes (fp32 and OpenCV's fp16 blobs); the folded / fused plan run by oracle/ssd_ref.py equals the UNFUSED
layer-by-layer interpreter oracle/caffe_ref.py on the same prototxt.  GPU: the HIP detector executing that plan
from the blob equals the plan's oracle tensor by tensor, boxes bit for bit."""
import numpy as np
import pytest
import torch

import caffe_net
import frames as F
from oracle import caffe_ref, ssd_ref


@pytest.fixture(scope="module")
def net():
    text, model, blobs = caffe_net.build(seed=3)
    return text, model, blobs


def test_parsers_round_trip(pkg, net):
    text, model, blobs = net
    C = pkg.caffe_io
    got = C.parse_caffemodel(model)
    assert set(got) == set(blobs)
    for k in blobs:
        assert len(got[k]) == len(blobs[k])
        for a, b in zip(got[k], blobs[k]):
            assert a.shape == b.shape and np.array_equal(a, b), k
    half = C.parse_caffemodel(caffe_net.write_caffemodel(blobs, fp16=True))       # OpenCV's fp16 caffemodel flavour
    for k in blobs:
        for a, b in zip(half[k], blobs[k]):
            assert np.array_equal(a, b.astype(np.float16).astype(np.float32)), k
    msg = C.parse_prototxt(text)
    layers = msg.getall("layer")
    assert msg.get("input") == "data" and [int(d) for d in msg.get("input_shape").getall("dim")] == [1, 3, 300, 300]
    assert layers[0].get("type") == "BatchNorm" and layers[-1].get("type") == "DetectionOutput"
    pb = [l for l in layers if l.get("type") == "PriorBox"][1].get("prior_box_param")
    assert pb.getall("aspect_ratio") == [2, 3] and pb.get("flip") is True and pb.getall("variance") == [0.1, 0.1, 0.2, 0.2]
    assert layers[-1].get("detection_output_param").get("code_type") == "CENTER_SIZE"


def test_plan_folds_and_fuses_the_graph(pkg, net):
    text, model, _ = net
    arch, sd = pkg.caffe_io.build_arch(text, pkg.caffe_io.parse_caffemodel(model))
    kinds = [k for _, k, _ in arch.LAYERS]
    names = [n for n, _, _ in arch.LAYERS]
    # no BatchNorm / Scale / ReLU / Eltwise layer survives on its own except the pre-activation affines
    assert set(kinds) == {"conv", "maxpool", "affine", "l2norm"}
    assert kinds.count("affine") == 4 and all(n.endswith("_bn1") or n == "last_bn_h" for n, k, _ in arch.LAYERS if k == "affine")
    by = {n: a for n, _, a in arch.LAYERS}
    assert by["layer_64_1_conv2_h"][8] == "conv1_pool"                            # Eltwise fused as the conv's residual
    assert by["layer_128_1_conv2_h"][8] == "layer_128_1_conv_expand_h"
    assert names.index("layer_128_1_conv_expand_h") < names.index("layer_128_1_conv2_h")   # operand emitted first
    assert by["conv1_h"][7] is True and by["layer_128_1_conv_expand_h"][7] is False
    assert arch.IN_SCALE != (1.0, 1.0, 1.0)                                       # the data BatchNorm went into the input transform
    assert [s[0] for s in arch.SOURCES] == ["conv4_3_norm", "last_bn_h", "conv6_2_h", "conv7_2_h", "conv8_2_h", "conv9_2_h"]
    assert [s[2] for s in arch.SOURCES] == [38, 19, 10, 5, 3, 1] and arch.TOP_K == 400 and arch.KEEP_TOP_K == 200


@pytest.mark.parametrize("frame", [F.natural_like(480, 640, seed=5), F.face_frame(640, 480, 1)], ids=["natural", "face"])
def test_fused_plan_equals_unfused_interpreter(pkg, net, frame):
    text, model, _ = net
    C = pkg.caffe_io
    blobs = C.parse_caffemodel(model)
    arch, sd = C.build_arch(text, blobs)
    want = caffe_ref.run(C.parse_prototxt(text), blobs, frame)
    got = ssd_ref.forward(pkg.weights.to_torch(sd), arch, frame)
    assert len(want) == len(got) and len(got) > 0
    assert np.abs(np.asarray(want) - np.asarray(got)).max() <= 2e-5
    assert ssd_ref.postprocess(got, 480, 640, 0.3) == ssd_ref.postprocess(want, 480, 640, 0.3)


@pytest.mark.gpu
def test_hip_detector_runs_the_imported_plan(pkg, net, seeded_sd):
    text, model, _ = net
    arch, sd = pkg.caffe_io.build_arch(text, pkg.caffe_io.parse_caffemodel(model))
    h = pkg._lib.Handle(pkg.weights.pack_all(seeded_sd, sd, None, ssd_arch=arch), device=0, max_batch=4)
    tsd = pkg.weights.to_torch(sd)
    try:
        assert h.has_detector
        for frame in (F.natural_like(480, 640, seed=5), F.face_frame(640, 480, 1), F.natural_like(720, 1280, seed=8)):
            taps = {}
            rows = ssd_ref.forward(tsd, arch, frame, taps)
            for name in ("conv1_h", "layer_64_1_conv2_h", "layer_128_1_bn1", "layer_256_1_conv2_h", "last_bn_h", "conv4_3_norm", "conv9_2_h"):
                w = taps[name].permute(0, 2, 3, 1).contiguous().numpy().reshape(-1)
                got = h.ssd_tap(frame, name, w.size)
                assert np.abs(got - w).max() <= 2e-4 * max(1.0, float(np.abs(w).max())), name
            hh, ww = frame.shape[:2]
            for thr in (0.3, 0.5):
                assert h.detect_faces(frame, thr) == ssd_ref.postprocess(rows, hh, ww, thr)
        assert len(ssd_ref.postprocess(rows, hh, ww, 0.3)) > 0
    finally:
        h.close()
