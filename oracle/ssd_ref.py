"""ORACLE (test infrastructure only): CPU restatement of the reference's SSD face-detection path.

Follows reference face_detection.py:37-105: the guards of `detect_bounding_box` (:51-68), then
`_detect_dnn`: cv2.resize to 300x300 -> blobFromImage(scale 1, mean BGR (104,177,123), no
swap) -> net.forward() -> rows (img, cls, conf, x1, y1, x2, y2) -> conf > thr (strict) ->
scale by [w,h,w,h] (a float32 * int64 numpy product, i.e. float64) -> astype(int) (truncate)
-> clamp -> keep bw > 20 and bh > 20 -> (x1, y1, bw, bh) in network order.

The network itself (res10-SSD through cv2.dnn) is NOT in the reference tree; the layer list is
taken as a parameter (the package's ssd_arch tables) and executed with torch-CPU fp32, with
Caffe's PriorBox and DetectionOutput semantics restated from their published definitions:
CENTER_SIZE decoding with variances, 2-way softmax, per-class stable sort by score, top_k,
greedy NMS (overlap > 0.45 suppresses), keep_top_k.  PARITY UNPINNED: the reference tests pin
only "returns a list and never crashes" for this path (tests/test_functional.py:117-157,
tests/test_reliability.py:27-38,104-113) - those guards ARE covered in tests/test_ssd_ref.py.
"""
from __future__ import annotations

import math
from typing import List, Mapping, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import imgproc_ref as I


def preprocess(frame: np.ndarray, size: int, mean_bgr, in_scale=None, in_shift=None) -> torch.Tensor:
    """cv2.resize + blobFromImage -> (1,3,size,size) float32, BGR order.  With in_scale / in_shift (an imported
    topology whose data blob passes a BatchNorm/Scale before the first convolution): x * scale + shift instead of
    x - mean (the shift already contains the mean)."""
    r = I.resize_linear_u8(frame, size, size).astype(np.float32)
    if in_scale is None:
        r -= np.asarray(mean_bgr, np.float32)
    else:
        r = r * np.asarray(in_scale, np.float32) + np.asarray(in_shift, np.float32)
    return torch.from_numpy(r).permute(2, 0, 1).unsqueeze(0).contiguous()


def _maxpool_ceil(x, k, s):
    return F.max_pool2d(x, k, s, 0, ceil_mode=True)


def run_trunk(sd: Mapping[str, torch.Tensor], layers, x: torch.Tensor, taps=None):
    t = {"data": x}
    for name, kind, a in layers:
        if kind == "conv":
            src, ci, co, k, s, p, d, relu, res = a
            y = F.conv2d(t[src], sd[name + ".weight"], sd[name + ".bias"], stride=s, padding=p, dilation=d)
            if res is not None:
                y = y + t[res]
            t[name] = F.relu(y) if relu else y
        elif kind == "maxpool":
            src, k, s = a
            t[name] = _maxpool_ceil(t[src], k, s)
        elif kind == "l2norm":
            src = a[0]
            v = t[src]
            t[name] = v / torch.sqrt((v * v).sum(1, keepdim=True) + 1e-10) * sd[name + ".scale"].view(1, -1, 1, 1)
        elif kind == "affine":
            src, c, relu = a
            y = t[src] * sd[name + ".scale"].view(1, -1, 1, 1) + sd[name + ".shift"].view(1, -1, 1, 1)
            t[name] = F.relu(y) if relu else y
        elif kind == "add":
            src, other, c, relu = a
            y = t[src] + t[other]
            t[name] = F.relu(y) if relu else y
        if taps is not None:
            taps[name] = t[name]
    return t


def prior_boxes(sources, image_size: int) -> np.ndarray:
    """Caffe PriorBox, clip=false, offset 0.5: rows (xmin,ymin,xmax,ymax) normalised, in the order
    source -> row -> column -> [min, sqrt(min*max), ar, 1/ar, ...]."""
    out = []
    for _, _, m, mn, mx, ars, step in sources:
        sizes = [(mn, mn), (math.sqrt(mn * mx), math.sqrt(mn * mx))]
        for ar in ars:
            r = math.sqrt(ar)
            sizes += [(mn * r, mn / r), (mn / r, mn * r)]
        for h in range(m):
            for w in range(m):
                cx, cy = (w + 0.5) * step, (h + 0.5) * step
                for bw, bh in sizes:
                    out.append(((cx - bw / 2.0) / image_size, (cy - bh / 2.0) / image_size,
                                (cx + bw / 2.0) / image_size, (cy + bh / 2.0) / image_size))
    return np.asarray(out, np.float32)


def heads(sd, sources, t) -> Tuple[np.ndarray, np.ndarray]:
    """-> loc (P,4), conf logits (P,2) in prior order (Caffe permutes each head to NHWC and flattens)."""
    locs, confs = [], []
    for src, c, m, _, _, ars, _ in sources:
        l = F.conv2d(t[src], sd[src + "_loc.weight"], sd[src + "_loc.bias"], padding=1)
        c_ = F.conv2d(t[src], sd[src + "_conf.weight"], sd[src + "_conf.bias"], padding=1)
        locs.append(l.permute(0, 2, 3, 1).reshape(-1, 4))
        confs.append(c_.permute(0, 2, 3, 1).reshape(-1, 2))
    return torch.cat(locs).numpy(), torch.cat(confs).numpy()


def decode(priors: np.ndarray, loc: np.ndarray, variances) -> np.ndarray:
    """CENTER_SIZE, variance_encoded_in_target = false; float32 throughout."""
    f = np.float32
    pw, ph = priors[:, 2] - priors[:, 0], priors[:, 3] - priors[:, 1]
    pcx, pcy = (priors[:, 0] + priors[:, 2]) * f(0.5), (priors[:, 1] + priors[:, 3]) * f(0.5)
    cx = f(variances[0]) * loc[:, 0] * pw + pcx
    cy = f(variances[1]) * loc[:, 1] * ph + pcy
    w = np.exp(f(variances[2]) * loc[:, 2]) * pw
    h = np.exp(f(variances[3]) * loc[:, 3]) * ph
    return np.stack([cx - w * f(0.5), cy - h * f(0.5), cx + w * f(0.5), cy + h * f(0.5)], 1).astype(np.float32)


def _area(b):
    return 0.0 if b[2] < b[0] or b[3] < b[1] else float(np.float32(b[2] - b[0]) * np.float32(b[3] - b[1]))


def jaccard(a, b) -> float:
    if b[0] > a[2] or b[2] < a[0] or b[1] > a[3] or b[3] < a[1]:
        return 0.0
    ix = np.float32(min(a[2], b[2])) - np.float32(max(a[0], b[0]))
    iy = np.float32(min(a[3], b[3])) - np.float32(max(a[1], b[1]))
    inter = np.float32(ix * iy)
    return float(inter / np.float32(np.float32(_area(a)) + np.float32(_area(b)) - inter))


def detection_output(boxes: np.ndarray, face_prob: np.ndarray, conf_thr, nms_thr, top_k, keep_top_k):
    """One image, one foreground class.  Returns rows (score, x1, y1, x2, y2) by descending score."""
    idx = [i for i in range(len(face_prob)) if face_prob[i] > conf_thr]
    idx.sort(key=lambda i: -face_prob[i])                      # Python's sort is stable, like std::stable_sort
    idx = idx[:top_k]
    keep: List[int] = []
    for i in idx:
        if all(jaccard(boxes[i], boxes[j]) <= nms_thr for j in keep):
            keep.append(i)
    keep = keep[:keep_top_k]
    return [(float(face_prob[i]),) + tuple(float(v) for v in boxes[i]) for i in keep]


@torch.no_grad()
def forward(sd, arch, frame: np.ndarray, taps=None):
    """frame (BGR u8) -> DetectionOutput rows.  `arch` is the package's ssd_arch module."""
    x = preprocess(frame, arch.INPUT, arch.MEAN_BGR, getattr(arch, "IN_SCALE", None), getattr(arch, "IN_SHIFT", None))
    t = run_trunk(sd, arch.LAYERS, x, taps)
    loc, conf = heads(sd, arch.SOURCES, t)
    e = np.exp(conf - conf.max(1, keepdims=True))
    prob = (e[:, 1] / e.sum(1)).astype(np.float32)
    boxes = decode(prior_boxes(arch.SOURCES, arch.INPUT), loc, arch.VARIANCES)
    if taps is not None:
        taps.update(loc=loc, conf=conf, prob=prob, boxes=boxes)
    return detection_output(boxes, prob, arch.CONF_THRESHOLD, arch.NMS_THRESHOLD, arch.TOP_K, arch.KEEP_TOP_K)


def postprocess(rows: Sequence[Sequence[float]], h: int, w: int, confidence_threshold=0.5):
    """reference face_detection.py:84-105 on DetectionOutput rows (score, x1, y1, x2, y2)."""
    out = []
    for r in rows:
        if np.float32(r[0]) > confidence_threshold:
            box = np.asarray(r[1:5], np.float32) * np.array([w, h, w, h])        # -> float64
            x1, y1, x2, y2 = box.astype("int")
            x1, y1, x2, y2 = max(0, x1), max(0, y1), min(w, x2), min(h, y2)
            bw, bh = x2 - x1, y2 - y1
            if bw > 20 and bh > 20:
                out.append((int(x1), int(y1), int(bw), int(bh)))
    return out


def detect_bounding_box(sd, arch, frame, confidence_threshold=0.5):
    """reference face_detection.py:37-68 (DNN branch; the Haar fallback is out of scope)."""
    try:
        if frame is None or frame.size == 0:
            return []
        if len(frame.shape) < 2 or frame.shape[0] < 30 or frame.shape[1] < 30:
            return []
        h, w = frame.shape[:2]
        return postprocess(forward(sd, arch, frame), h, w, confidence_threshold)
    except Exception:
        return []
