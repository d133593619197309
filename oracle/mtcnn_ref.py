"""CPU oracle for the MTCNN align/crop stage.  TEST INFRASTRUCTURE ONLY.

The reference calls ``mtcnn(PIL_image)`` on the already-cropped face
(reference deepfake_detection.py:24-28, 376-380) with ``MTCNN(select_largest=False,
post_process=False)``; the arithmetic lives in the third-party package facenet-pytorch
(requirements.txt:3, ``>=2.5.2``), which is ABSENT from /root/reference and from this image.
This file restates the package's published algorithm (models/mtcnn.py: PNet/RNet/ONet,
MTCNN.forward/detect/select_boxes/extract; models/utils/detect_face.py: detect_face,
generateBoundingBox, bbreg, rerec, pad, nms_numpy, batched_nms_numpy, imresample, extract_face,
crop_resize) with its defaults: image_size 160, margin 0, min_face_size 20, thresholds
(0.6, 0.7, 0.7), factor 0.709, selection by highest probability.

PARITY UNPINNED for the network cascade: the reference's tests hold no vector for it and the
package cannot be run here.  Pinned exactly: `pil_resize_bilinear` (the 8-bit fixed-point
``Image.resize(..., BILINEAR)`` behind crop_resize) against Pillow (tests/test_mtcnn_ref.py).
torchvision's ``batched_nms`` (also absent) is restated as greedy IoU suppression in descending
score order.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MIN_FACE, FACTOR = 20, 0.709
THRESHOLDS = (0.6, 0.7, 0.7)
IMAGE_SIZE = 160

# ------------------------------------------------------------------ PIL 8-bit bilinear resize
_PB = 22        # PRECISION_BITS = 32 - 8 - 2 (Pillow src/libImaging/Resample.c)


def pil_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(np.ceil(support)) * 2 + 1
    k = np.zeros((out_size, ksize), np.float64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / fs
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k[xx, x] = w
            ww += w
        if ww != 0.0:
            k[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    ki = np.where(k < 0, (-0.5 + k * (1 << _PB)).astype(np.int64), (0.5 + k * (1 << _PB)).astype(np.int64))
    return ki, bounds


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    ki, bounds = pil_coeffs(a.shape[0], out_size)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        ss = (1 << (_PB - 1)) + np.tensordot(ki[xx, :n], a[xmin:xmin + n], axes=(0, 0))
        out[xx] = np.clip(ss >> _PB, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR)``: horizontal pass, then vertical,
    8-bit intermediate (a pass is skipped when that size already matches)."""
    h, w = img.shape[:2]
    tmp = _resample_axis(img, out_w, 1) if out_w != w else img
    return _resample_axis(tmp, out_h, 0) if out_h != h else tmp


# ------------------------------------------------------------------------------- networks
def pnet(sd, x):
    x = F.prelu(F.conv2d(x, sd["pnet.conv1.weight"], sd["pnet.conv1.bias"]), sd["pnet.prelu1.weight"])
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, sd["pnet.conv2.weight"], sd["pnet.conv2.bias"]), sd["pnet.prelu2.weight"])
    x = F.prelu(F.conv2d(x, sd["pnet.conv3.weight"], sd["pnet.conv3.bias"]), sd["pnet.prelu3.weight"])
    a = torch.softmax(F.conv2d(x, sd["pnet.conv4_1.weight"], sd["pnet.conv4_1.bias"]), dim=1)
    b = F.conv2d(x, sd["pnet.conv4_2.weight"], sd["pnet.conv4_2.bias"])
    return b, a


def rnet(sd, x):
    x = F.prelu(F.conv2d(x, sd["rnet.conv1.weight"], sd["rnet.conv1.bias"]), sd["rnet.prelu1.weight"])
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, sd["rnet.conv2.weight"], sd["rnet.conv2.bias"]), sd["rnet.prelu2.weight"])
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, sd["rnet.conv3.weight"], sd["rnet.conv3.bias"]), sd["rnet.prelu3.weight"])
    x = x.permute(0, 3, 2, 1).contiguous()
    x = F.prelu(F.linear(x.view(x.shape[0], -1), sd["rnet.dense4.weight"], sd["rnet.dense4.bias"]), sd["rnet.prelu4.weight"])
    a = torch.softmax(F.linear(x, sd["rnet.dense5_1.weight"], sd["rnet.dense5_1.bias"]), dim=1)
    b = F.linear(x, sd["rnet.dense5_2.weight"], sd["rnet.dense5_2.bias"])
    return b, a


def onet(sd, x):
    x = F.prelu(F.conv2d(x, sd["onet.conv1.weight"], sd["onet.conv1.bias"]), sd["onet.prelu1.weight"])
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, sd["onet.conv2.weight"], sd["onet.conv2.bias"]), sd["onet.prelu2.weight"])
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, sd["onet.conv3.weight"], sd["onet.conv3.bias"]), sd["onet.prelu3.weight"])
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    x = F.prelu(F.conv2d(x, sd["onet.conv4.weight"], sd["onet.conv4.bias"]), sd["onet.prelu4.weight"])
    x = x.permute(0, 3, 2, 1).contiguous()
    x = F.prelu(F.linear(x.view(x.shape[0], -1), sd["onet.dense5.weight"], sd["onet.dense5.bias"]), sd["onet.prelu5.weight"])
    a = torch.softmax(F.linear(x, sd["onet.dense6_1.weight"], sd["onet.dense6_1.bias"]), dim=1)
    b = F.linear(x, sd["onet.dense6_2.weight"], sd["onet.dense6_2.bias"])
    c = F.linear(x, sd["onet.dense6_3.weight"], sd["onet.dense6_3.bias"])
    return b, c, a


# ------------------------------------------------------------------------------ box helpers
def scale_pyramid(h: int, w: int):
    """detect_face: m = 12 / minsize; scales m * factor^i while min(h, w) * scale >= 12 (double arithmetic)."""
    m = 12.0 / MIN_FACE
    minl = min(h, w) * m
    scale_i = m
    scales = []
    while minl >= 12:
        scales.append(scale_i)
        scale_i = scale_i * FACTOR
        minl = minl * FACTOR
    return scales


def imresample(img: torch.Tensor, size):
    """``interpolate(img, size=size, mode="area")`` = adaptive average pooling."""
    return F.interpolate(img, size=size, mode="area")


def generate_bounding_box(reg: torch.Tensor, probs: torch.Tensor, scale: float, thresh: float) -> np.ndarray:
    """rows (x1, y1, x2, y2, score, 4 x reg) of the cells with prob >= thresh, in (y, x) order."""
    stride, cellsize = 2, 12
    reg = reg[0].numpy()                 # (4, H, W)
    p = probs[0].numpy()                 # (H, W)
    ys, xs = np.nonzero(p >= np.float32(thresh))
    score = p[ys, xs]
    r = reg[:, ys, xs].T
    bb = np.stack([xs, ys], axis=1).astype(np.float32)
    q1 = np.floor((np.float32(stride) * bb + np.float32(1)) / np.float32(scale))
    q2 = np.floor((np.float32(stride) * bb + np.float32(cellsize - 1 + 1)) / np.float32(scale))
    return np.concatenate([q1, q2, score[:, None], r], axis=1).astype(np.float32)


def nms_iou(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """torchvision.ops.nms: descending score, suppress IoU > thr, areas without the +1."""
    if len(boxes) == 0:
        return np.zeros((0,), np.int64)
    b = boxes.astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep = []
    dead = np.zeros(len(b), bool)
    for oi, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
        iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > np.float32(thr)]] = True
    return np.asarray(keep, np.int64)


def nms_min(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """nms_numpy(..., method='Min'): +1 areas, overlap / min(area); keeps o <= thr."""
    if len(boxes) == 0:
        return np.zeros((0,), np.int64)
    b = boxes.astype(np.float32)
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    idx = np.argsort(scores, kind="stable")
    pick = []
    while idx.size > 0:
        i = idx[-1]
        pick.append(i)
        rest = idx[:-1]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        w = np.maximum(np.float32(0), xx2 - xx1 + 1); h = np.maximum(np.float32(0), yy2 - yy1 + 1)
        o = (w * h) / np.minimum(area[i], area[rest])
        idx = rest[o <= np.float32(thr)]
    return np.asarray(pick, np.int64)


def bbreg(bb: np.ndarray, reg: np.ndarray) -> np.ndarray:
    w = bb[:, 2] - bb[:, 0] + 1
    h = bb[:, 3] - bb[:, 1] + 1
    out = bb.copy()
    out[:, 0] = bb[:, 0] + reg[:, 0] * w
    out[:, 1] = bb[:, 1] + reg[:, 1] * h
    out[:, 2] = bb[:, 2] + reg[:, 2] * w
    out[:, 3] = bb[:, 3] + reg[:, 3] * h
    return out


def rerec(bb: np.ndarray) -> np.ndarray:
    h = bb[:, 3] - bb[:, 1]
    w = bb[:, 2] - bb[:, 0]
    l = np.maximum(w, h)
    out = bb.copy()
    out[:, 0] = bb[:, 0] + w * np.float32(0.5) - l * np.float32(0.5)
    out[:, 1] = bb[:, 1] + h * np.float32(0.5) - l * np.float32(0.5)
    out[:, 2] = out[:, 0] + l
    out[:, 3] = out[:, 1] + l
    return out


def pad(bb: np.ndarray, w: int, h: int):
    b = np.trunc(bb[:, :4]).astype(np.int32)
    x, y, ex, ey = b[:, 0].copy(), b[:, 1].copy(), b[:, 2].copy(), b[:, 3].copy()
    x[x < 1] = 1
    y[y < 1] = 1
    ex[ex > w] = w
    ey[ey > h] = h
    return y, ey, x, ex


def _crops(img: torch.Tensor, bb: np.ndarray, size: int):
    """second/third stage input: img[:, y-1:ey, x-1:ex] area-resampled to size^2, normalised; boxes whose
    slice is empty are dropped from the batch (detect_face keeps their rows: such a batch would then
    mis-align in the package; the oracle reports them through `valid`)."""
    h, w = img.shape[2:]
    y, ey, x, ex = pad(bb, w, h)
    out, valid = [], []
    for k in range(len(y)):
        ok = ey[k] > (y[k] - 1) and ex[k] > (x[k] - 1)
        valid.append(ok)
        if ok:
            out.append(imresample(img[:, :, y[k] - 1:ey[k], x[k] - 1:ex[k]], (size, size)))
    if not out:
        return torch.zeros(0, 3, size, size), np.asarray(valid, bool)
    return (torch.cat(out, 0) - 127.5) * 0.0078125, np.asarray(valid, bool)


def detect_face(sd, rgb: np.ndarray, taps: dict | None = None) -> np.ndarray:
    """rgb uint8 (H, W, 3) -> rows (x1, y1, x2, y2, prob) after the three stages (float32)."""
    with torch.no_grad():
        img = torch.from_numpy(np.ascontiguousarray(rgb)).permute(2, 0, 1)[None].float()
        h, w = img.shape[2:]
        rows = []
        for si, scale in enumerate(scale_pyramid(h, w)):
            data = (imresample(img, (int(h * scale + 1), int(w * scale + 1))) - 127.5) * 0.0078125
            reg, probs = pnet(sd, data)
            if taps is not None:
                taps[f"pnet.prob.{si}"] = probs[0, 1].numpy().copy()
                taps[f"pnet.reg.{si}"] = reg[0].numpy().copy()
            bs = generate_bounding_box(reg, probs[:, 1], scale, THRESHOLDS[0])
            rows.append(bs[nms_iou(bs[:, :4], bs[:, 4], 0.5)])
        boxes = np.concatenate(rows, 0) if rows else np.zeros((0, 9), np.float32)
        boxes = boxes[nms_iou(boxes[:, :4], boxes[:, 4], 0.7)]
        regw = boxes[:, 2] - boxes[:, 0]
        regh = boxes[:, 3] - boxes[:, 1]
        boxes = np.stack([boxes[:, 0] + boxes[:, 5] * regw, boxes[:, 1] + boxes[:, 6] * regh,
                          boxes[:, 2] + boxes[:, 7] * regw, boxes[:, 3] + boxes[:, 8] * regh, boxes[:, 4]], 1)
        boxes = rerec(boxes.astype(np.float32))
        if taps is not None:
            taps["stage1"] = boxes.copy()
        if len(boxes):
            data, valid = _crops(img, boxes, 24)
            boxes = boxes[valid]
            reg, prob = rnet(sd, data)
            score = prob[:, 1].numpy()
            if taps is not None:
                taps["rnet.prob"] = score.copy()
            ipass = score > np.float32(THRESHOLDS[1])
            boxes = np.concatenate([boxes[ipass, :4], score[ipass, None]], 1)
            mv = reg.numpy()[ipass]
            pick = nms_iou(boxes[:, :4], boxes[:, 4], 0.7)
            boxes = rerec(bbreg(boxes[pick], mv[pick]))
        if taps is not None:
            taps["stage2"] = boxes.copy()
        if len(boxes):
            data, valid = _crops(img, boxes, 48)
            boxes = boxes[valid]
            reg, _points, prob = onet(sd, data)
            score = prob[:, 1].numpy()
            if taps is not None:
                taps["onet.prob"] = score.copy()
            ipass = score > np.float32(THRESHOLDS[2])
            boxes = np.concatenate([boxes[ipass, :4], score[ipass, None]], 1)
            boxes = bbreg(boxes, reg.numpy()[ipass])
            boxes = boxes[nms_min(boxes[:, :4], boxes[:, 4], 0.7)]
        if taps is not None:
            taps["stage3"] = boxes.copy()
        return boxes.astype(np.float32)


def extract_face(rgb: np.ndarray, box) -> np.ndarray:
    """extract_face(img, box, 160, margin=0): integer box clipped to the image, PIL crop + BILINEAR resize."""
    h, w = rgb.shape[:2]
    x1, y1 = int(max(box[0], 0)), int(max(box[1], 0))
    x2, y2 = int(min(box[2], w)), int(min(box[3], h))
    if x2 <= x1 or y2 <= y1:
        raise ValueError("degenerate face box")      # PIL raises here; the reference call site returns None
    return pil_resize_bilinear(rgb[y1:y2, x1:x2], IMAGE_SIZE, IMAGE_SIZE)


def mtcnn_forward(sd, rgb: np.ndarray, taps: dict | None = None):
    """MTCNN.forward for one image, keep_all=False, selection by probability, post_process=False:
    (3, 160, 160) float32 RGB in 0..255, or None when no face passes."""
    boxes = detect_face(sd, rgb, taps)
    if len(boxes) == 0:
        return None
    best = boxes[np.argsort(boxes[:, 4], kind="stable")[::-1][0]]
    if taps is not None:
        taps["selected"] = best.copy()
    try:
        face = extract_face(rgb, best[:4])
    except ValueError:
        return None
    return np.ascontiguousarray(face.transpose(2, 0, 1)).astype(np.float32)
