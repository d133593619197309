"""Topology of the SSD-style face detector (host-side constants).

The reference runs OpenCV's ``res10_300x300_ssd`` Caffe model through ``cv2.dnn``
(reference face_detection.py:19-24,71-82); neither the prototxt nor the caffemodel is in the
reference tree, so the layer list below is this build's own statement of that
family: a ResNet-10-style trunk on a 300x300 input, SSD extra layers, six source maps
(38,19,10,5,3,1) with 4,6,6,6,4,4 priors = 8732 priors, two classes, Caffe PriorBox /
DetectionOutput semantics (min/max sizes 30..315, variances .1 .1 .2 .2, NMS 0.45, top_k 400,
keep_top_k 200, confidence 0.01).  BatchNorm/Scale are folded into the convolutions, as any
inference deployment of the Caffe model does.  Loading the real Caffe weights into this layer
list is the "next" row N3 of SURVEY.md section 8(f).

A layer is (name, kind, args); tensors are named by the layer that produces them.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

INPUT = 300
MEAN_BGR = (104.0, 177.0, 123.0)          # reference face_detection.py:78
# the first convolution sees x * IN_SCALE + IN_SHIFT per channel (blobFromImage's mean subtraction; a prototxt with a
# BatchNorm/Scale on the data blob folds into these two, caffe_io.build_arch)
IN_SCALE = (1.0, 1.0, 1.0)
IN_SHIFT = (-104.0, -177.0, -123.0)
NUM_CLASSES = 2
VARIANCES = (0.1, 0.1, 0.2, 0.2)
NMS_THRESHOLD = 0.45
TOP_K = 400
KEEP_TOP_K = 200
CONF_THRESHOLD = 0.01
NORM_SCALE_INIT = 20.0

# conv: (src, c_in, c_out, k, stride, pad, dilation, relu, residual_src or None)
#   residual is added BEFORE the ReLU (ResNet basic block)
# further kinds an imported topology may use (caffe_io): "affine" (src, c, relu): y = [relu](x * scale[c] + shift[c]),
#   a BatchNorm+Scale that cannot be folded into a convolution; "add" (src, other, c, relu): Eltwise SUM
LAYERS: List[Tuple[str, str, tuple]] = [
    ("conv1", "conv", ("data", 3, 32, 7, 2, 3, 1, True, None)),               # 150
    ("pool1", "maxpool", ("conv1", 3, 2)),                                     # 75 (ceil mode)
    ("res2a", "conv", ("pool1", 32, 32, 3, 1, 1, 1, True, None)),
    ("res2b", "conv", ("res2a", 32, 32, 3, 1, 1, 1, True, "pool1")),
    ("res3p", "conv", ("res2b", 32, 128, 1, 2, 0, 1, False, None)),            # projection shortcut, 38
    ("res3a", "conv", ("res2b", 32, 128, 3, 2, 1, 1, True, None)),
    ("res3b", "conv", ("res3a", 128, 128, 3, 1, 1, 1, True, "res3p")),
    ("res4p", "conv", ("res3b", 128, 256, 1, 2, 0, 1, False, None)),           # 19
    ("res4a", "conv", ("res3b", 128, 256, 3, 2, 1, 1, True, None)),
    ("res4b", "conv", ("res4a", 256, 256, 3, 1, 1, 1, True, "res4p")),
    ("res5a", "conv", ("res4b", 256, 256, 3, 1, 2, 2, True, None)),            # dilated, 19
    ("res5b", "conv", ("res5a", 256, 256, 3, 1, 2, 2, True, "res4b")),
    ("conv6_1", "conv", ("res5b", 256, 128, 1, 1, 0, 1, True, None)),
    ("conv6_2", "conv", ("conv6_1", 128, 256, 3, 2, 1, 1, True, None)),        # 10
    ("conv7_1", "conv", ("conv6_2", 256, 64, 1, 1, 0, 1, True, None)),
    ("conv7_2", "conv", ("conv7_1", 64, 128, 3, 2, 1, 1, True, None)),         # 5
    ("conv8_1", "conv", ("conv7_2", 128, 64, 1, 1, 0, 1, True, None)),
    ("conv8_2", "conv", ("conv8_1", 64, 128, 3, 1, 0, 1, True, None)),         # 3
    ("conv9_1", "conv", ("conv8_2", 128, 64, 1, 1, 0, 1, True, None)),
    ("conv9_2", "conv", ("conv9_1", 64, 128, 3, 1, 0, 1, True, None)),         # 1
    ("norm3", "l2norm", ("res3b", 128)),                                       # SSD Normalize on the first source
]

# source maps: (tensor, channels, map size, min_size, max_size, aspect ratios, step)
SOURCES = [
    ("norm3", 128, 38, 30.0, 60.0, (2.0,), 8.0),
    ("res5b", 256, 19, 60.0, 111.0, (2.0, 3.0), 16.0),
    ("conv6_2", 256, 10, 111.0, 162.0, (2.0, 3.0), 32.0),
    ("conv7_2", 128, 5, 162.0, 213.0, (2.0, 3.0), 64.0),
    ("conv8_2", 128, 3, 213.0, 264.0, (2.0,), 100.0),
    ("conv9_2", 128, 1, 264.0, 315.0, (2.0,), 300.0),
]


def priors_per_cell(ars) -> int:
    return 2 + 2 * len(ars)


def num_priors() -> int:
    return sum(s[2] * s[2] * priors_per_cell(s[5]) for s in SOURCES)


def conv_out(size: int, k: int, stride: int, pad: int, dil: int) -> int:
    return (size + 2 * pad - dil * (k - 1) - 1) // stride + 1


def pool_out_ceil(size: int, k: int, stride: int) -> int:
    return -(-(size - k) // stride) + 1


def shapes() -> Dict[str, Tuple[int, int]]:
    """tensor name -> (channels, spatial size)"""
    out = {"data": (3, INPUT)}
    for name, kind, a in LAYERS:
        if kind == "conv":
            src, ci, co, k, s, p, d, _, _ = a
            out[name] = (co, conv_out(out[src][1], k, s, p, d))
        elif kind == "maxpool":
            src, k, s = a
            out[name] = (out[src][0], pool_out_ceil(out[src][1], k, s))
        elif kind == "l2norm":
            out[name] = out[a[0]]
    return out


def macs_per_frame() -> int:
    sh = shapes()
    n = 0
    for name, kind, a in LAYERS:
        if kind == "conv":
            src, ci, co, k, s, p, d, _, _ = a
            n += sh[name][1] ** 2 * k * k * ci * co
    for t, c, m, _, _, ars, _ in SOURCES:
        n += m * m * 9 * c * priors_per_cell(ars) * 6
    return n
