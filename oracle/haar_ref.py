"""ORACLE (test infrastructure only): cv2.CascadeClassifier.detectMultiScale for a stump cascade of upright HAAR
features, restated on numpy (reference face_detection.py:108-123 calls it with scaleFactor 1.1, minNeighbors 5,
minSize (30, 30) on the BGR2GRAY image).  Follows OpenCV's objdetect/cascadedetect.cpp as published: the scale loop
(window = round(win * factor), image = round(size / factor), INTER_LINEAR resize), HaarEvaluator::setWindow's variance
normalisation over the window shrunk by one pixel, predictOrderedStump, the "rejected by stage 0 -> skip the next
position" rule of CascadeClassifierInvoker, and groupRectangles(eps 0.2).

PARITY UNPINNED: cv2 and its cascade XML are absent here; the float evaluation order (float feature sums, float
norm, double stage sums) is this restatement's reading of that source."""
from __future__ import annotations

import numpy as np

from . import imgproc_ref as I


def _round(v):
    return int(np.rint(v))                                    # cvRound: half to even


def _integral(img):
    s = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int64)
    s[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.int64), 0), 1)
    return s


def _rect(s, x0, y0, w, h, ys, xs):
    """rect sum for every window origin (ys[:, None], xs[None, :])"""
    a = s[np.ix_(ys + y0, xs + x0)]
    b = s[np.ix_(ys + y0, xs + x0 + w)]
    c = s[np.ix_(ys + y0 + h, xs + x0)]
    d = s[np.ix_(ys + y0 + h, xs + x0 + w)]
    return a - b - c + d


def candidates(gray: np.ndarray, cas: dict, scale_factor=1.1, min_size=30):
    H, W = gray.shape
    win_w, win_h = int(cas["haar.win"][0]), int(cas["haar.win"][1])
    stages, stumps, rects = cas["haar.stages"], cas["haar.stumps"], cas["haar.rects"].reshape(-1, 3, 5)
    out = []
    factor = 1.0
    while True:
        ww, wh = _round(win_w * factor), _round(win_h * factor)
        sw, sh = _round(W / factor), _round(H / factor)
        if sw - win_w <= 0 or sh - win_h <= 0:
            break
        if ww >= min_size and wh >= min_size:
            img = gray if (sw, sh) == (W, H) else I.resize_linear_u8(gray[..., None], sw, sh)[..., 0]
            s, q = _integral(img), _integral(img.astype(np.int64) ** 2)
            step = 1 if factor > 2.0 else 2
            ys, xs = np.arange(0, sh - win_h, step), np.arange(0, sw - win_w, step)
            nw, nh = win_w - 2, win_h - 2
            vs = _rect(s, 1, 1, nw, nh, ys, xs).astype(np.float64)
            vq = _rect(q, 1, 1, nw, nh, ys, xs).astype(np.float64)
            nf = float(nw * nh) * vq - vs * vs
            norm = np.where(nf > 0, 1.0 / np.sqrt(np.where(nf > 0, nf, 1.0)), 1.0).astype(np.float32)
            result = np.ones(vs.shape, np.int64)
            alive = np.ones(vs.shape, bool)
            for si, (first, cnt, thr) in enumerate(stages):
                acc = np.zeros(vs.shape, np.float64)
                for k in range(int(first), int(first + cnt)):
                    fi, th, left, right = stumps[k]
                    r = rects[int(fi)]
                    v = np.float32(r[0, 4]) * _rect(s, *(int(t) for t in r[0, :4]), ys, xs).astype(np.float32) + \
                        np.float32(r[1, 4]) * _rect(s, *(int(t) for t in r[1, :4]), ys, xs).astype(np.float32)
                    if r[2, 4] != 0:
                        v = v + np.float32(r[2, 4]) * _rect(s, *(int(t) for t in r[2, :4]), ys, xs).astype(np.float32)
                    acc += np.where((v * norm) < np.float32(th), np.float32(left), np.float32(right)).astype(np.float64)
                rej = alive & (acc < np.float64(np.float32(thr)))
                result[rej] = -si
                alive &= ~rej
            for gy in range(len(ys)):                          # the sequential skip rule along a row
                gx = 0
                while gx < len(xs):
                    r = result[gy, gx]
                    if r > 0:
                        out.append((_round(xs[gx] * factor), _round(ys[gy] * factor), ww, wh))
                    if r == 0:
                        gx += 1
                    gx += 1
        factor *= scale_factor
    return out


def group_rectangles(rects, group_threshold=5, eps=0.2):
    n = len(rects)
    if group_threshold <= 0 or n == 0:
        return list(rects)
    parent = list(range(n))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i

    def similar(a, b):
        delta = eps * (min(a[2], b[2]) + min(a[3], b[3])) * 0.5
        return (abs(a[0] - b[0]) <= delta and abs(a[1] - b[1]) <= delta and abs(a[0] + a[2] - b[0] - b[2]) <= delta
                and abs(a[1] + a[3] - b[1] - b[3]) <= delta)

    for i in range(n):
        for j in range(i + 1, n):
            if similar(rects[i], rects[j]):
                a, b = find(i), find(j)
                if a != b:
                    parent[b] = a
    label, cls = {}, []
    for i in range(n):
        r = find(i)
        if r not in label:
            label[r] = len(label)
        cls.append(label[r])
    k = len(label)
    sums = np.zeros((k, 4), np.int64)
    cnt = np.zeros(k, np.int64)
    for i, c in enumerate(cls):
        sums[c] += rects[i]
        cnt[c] += 1
    mean = [tuple(_round(np.float32(v) * (np.float32(1.0) / np.float32(cnt[c]))) for v in sums[c]) for c in range(k)]
    out = []
    for i in range(k):
        r1, n1 = mean[i], cnt[i]
        if n1 <= group_threshold:
            continue
        keep = True
        for j in range(k):
            n2 = cnt[j]
            if j == i or n2 <= group_threshold:
                continue
            r2 = mean[j]
            dx, dy = _round(r2[2] * eps), _round(r2[3] * eps)
            if (r1[0] >= r2[0] - dx and r1[1] >= r2[1] - dy and r1[0] + r1[2] <= r2[0] + r2[2] + dx
                    and r1[1] + r1[3] <= r2[1] + r2[3] + dy and (n2 > max(3, n1) or n1 < 3)):
                keep = False
                break
        if keep:
            out.append(r1)
    return out


def detect(frame_bgr: np.ndarray, cas: dict, scale_factor=1.1, min_neighbors=5, min_size=30):
    """-> (boxes [(x, y, w, h)], number of candidate windows)"""
    gray = I.bgr2gray_u8(frame_bgr) if frame_bgr.ndim == 3 else frame_bgr
    c = candidates(gray, cas, scale_factor, min_size)
    return group_rectangles(c, min_neighbors, 0.2), len(c)
