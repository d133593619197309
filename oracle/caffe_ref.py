"""ORACLE (test infrastructure only): a layer-by-layer interpreter of a Caffe SSD deploy.prototxt on torch-CPU,
UNFUSED - every BatchNorm, Scale, ReLU, Eltwise, Permute, Flatten, Concat, PriorBox and the DetectionOutput layer
are executed as Caffe defines them.  It is the check for `caffe_io.build_arch`, which folds and
fuses the same graph into the detector plan: both must produce the same DetectionOutput rows.

What it restates: Caffe's layer semantics as published in caffe.proto / the SSD fork's layers (reference
face_detection.py:19-24 loads such a net through cv2.dnn, whose Caffe importer implements the same semantics).
PARITY UNPINNED: no Caffe or cv2 here to run a real deploy.prototxt against; the nets in the tests are written by
the tests in the public res10 naming style.

Input: the parsed prototxt (`caffe_io.parse_prototxt` Msg - the text parser is exercised by its own tests) and the
blobs dict (layer name -> list of arrays)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ssd_ref


def _rep(msg, key, default):
    v = msg.getall(key)
    return v if v else [default]


def prior_box(pm, fm_h, fm_w, img_h, img_w):
    """caffe/layers/prior_box_layer.cpp Forward_cpu -> (boxes [P,4], variances [4])"""
    mins = [float(v) for v in pm.getall("min_size")]
    maxs = [float(v) for v in pm.getall("max_size")]
    flip = bool(pm.get("flip", True))
    ars = [1.0]
    for a in (float(v) for v in pm.getall("aspect_ratio")):
        if all(abs(a - e) > 1e-6 for e in ars):
            ars.append(a)
            if flip:
                ars.append(1.0 / a)
    step = float(pm.get("step")) if "step" in pm else None
    step_w = step if step else img_w / fm_w
    step_h = step if step else img_h / fm_h
    off = float(pm.get("offset", 0.5))
    out = []
    for h in range(fm_h):
        for w in range(fm_w):
            cx, cy = (w + off) * step_w, (h + off) * step_h
            for s, mn in enumerate(mins):
                bw = bh = mn
                out.append(((cx - bw / 2) / img_w, (cy - bh / 2) / img_h, (cx + bw / 2) / img_w, (cy + bh / 2) / img_h))
                if maxs:
                    bw = bh = math.sqrt(mn * maxs[s])
                    out.append(((cx - bw / 2) / img_w, (cy - bh / 2) / img_h, (cx + bw / 2) / img_w, (cy + bh / 2) / img_h))
                for a in ars:
                    if abs(a - 1.0) < 1e-6:
                        continue
                    bw, bh = mn * math.sqrt(a), mn / math.sqrt(a)
                    out.append(((cx - bw / 2) / img_w, (cy - bh / 2) / img_h, (cx + bw / 2) / img_w, (cy + bh / 2) / img_h))
    boxes = np.asarray(out, np.float32)
    if bool(pm.get("clip", False)):
        boxes = np.clip(boxes, 0.0, 1.0)
    var = [float(v) for v in pm.getall("variance")] or [0.1]
    return boxes, (var if len(var) == 4 else [var[0]] * 4)


def run(net, blobs, frame_bgr: np.ndarray, mean_bgr=(104.0, 177.0, 123.0)):
    """frame -> DetectionOutput rows [(score, x1, y1, x2, y2)] (class 1), exactly as ssd_ref.forward returns them"""
    layers = net.getall("layer")
    dims = None
    if "input" in net:
        shp = net.get("input_shape")
        dims = [int(d) for d in shp.getall("dim")] if shp is not None else [int(d) for d in net.getall("input_dim")]
        in_name = net.get("input")
    t = {}
    priors, variances = {}, None
    for L in layers:
        name, typ = L.get("name"), L.get("type")
        bot, top = L.getall("bottom"), L.getall("top")
        if typ == "Input":
            dims = [int(d) for d in L.get("input_param").get("shape").getall("dim")]
            in_name = top[0]
            continue
        if in_name not in t:
            t[in_name] = ssd_ref.preprocess(frame_bgr, dims[2], mean_bgr)
        x = t[bot[0]] if bot else None
        if typ == "Convolution":
            p = L.get("convolution_param")
            k = int(_rep(p, "kernel_size", 1)[0])
            w = torch.from_numpy(np.asarray(blobs[name][0], np.float32).reshape(int(p.get("num_output")), -1, k, k))
            b = torch.from_numpy(np.asarray(blobs[name][1], np.float32).reshape(-1)) if bool(p.get("bias_term", True)) else None
            y = F.conv2d(x, w, b, stride=int(_rep(p, "stride", 1)[0]), padding=int(_rep(p, "pad", 0)[0]),
                         dilation=int(_rep(p, "dilation", 1)[0]))
        elif typ == "BatchNorm":
            m = torch.from_numpy(np.asarray(blobs[name][0], np.float32).reshape(-1))
            v = torch.from_numpy(np.asarray(blobs[name][1], np.float32).reshape(-1))
            sf = float(np.asarray(blobs[name][2]).reshape(-1)[0]) if len(blobs[name]) > 2 else 1.0
            sf = 1.0 / sf if sf != 0 else 0.0
            eps = float(L.get("batch_norm_param").get("eps", 1e-5)) if "batch_norm_param" in L else 1e-5
            y = (x - (m * sf).view(1, -1, 1, 1)) / torch.sqrt((v * sf).view(1, -1, 1, 1) + eps)
        elif typ == "Scale":
            g = torch.from_numpy(np.asarray(blobs[name][0], np.float32).reshape(-1)).view(1, -1, 1, 1)
            y = x * g
            if "scale_param" in L and bool(L.get("scale_param").get("bias_term", False)):
                y = y + torch.from_numpy(np.asarray(blobs[name][1], np.float32).reshape(-1)).view(1, -1, 1, 1)
        elif typ == "ReLU":
            y = F.relu(x)
        elif typ == "Pooling":
            p = L.get("pooling_param")
            y = F.max_pool2d(x, int(p.get("kernel_size")), int(p.get("stride", 1)), int(p.get("pad", 0)), ceil_mode=True)
        elif typ == "Eltwise":
            y = t[bot[0]] + t[bot[1]]
        elif typ == "Normalize":
            s = torch.from_numpy(np.asarray(blobs[name][0], np.float32).reshape(-1)).view(1, -1, 1, 1)
            y = x / torch.sqrt((x * x).sum(1, keepdim=True) + 1e-10) * s
        elif typ == "Permute":
            y = x.permute(*[int(o) for o in L.get("permute_param").getall("order")]).contiguous()
        elif typ == "Flatten":
            ax = int(L.get("flatten_param").get("axis", 1)) if "flatten_param" in L else 1
            y = x.reshape(*x.shape[:ax], -1)
        elif typ == "Reshape":
            dimsr = [int(d) for d in L.get("reshape_param").get("shape").getall("dim")]
            shape = [x.shape[i] if d == 0 else d for i, d in enumerate(dimsr)]
            y = x.reshape(*shape)
        elif typ == "Softmax":
            ax = int(L.get("softmax_param").get("axis", 1)) if "softmax_param" in L else 1
            y = torch.softmax(x, ax)
        elif typ == "Concat":
            ax = int(L.get("concat_param").get("axis", 1)) if "concat_param" in L else 1
            if all(b in priors for b in bot):
                priors[top[0]] = np.concatenate([priors[b] for b in bot])
                t[top[0]] = None
                continue
            y = torch.cat([t[b] for b in bot], ax)
        elif typ == "PriorBox":
            boxes, variances = prior_box(L.get("prior_box_param"), x.shape[2], x.shape[3], dims[2], dims[3])
            priors[top[0]] = boxes
            t[top[0]] = None
            continue
        elif typ == "DetectionOutput":
            dp = L.get("detection_output_param")
            loc = t[bot[0]].reshape(-1, 4).numpy()
            prob = t[bot[1]].reshape(-1, 2).numpy()[:, 1].astype(np.float32)
            pri = priors[bot[2]]
            boxes = ssd_ref.decode(pri, loc, variances)
            nms = dp.get("nms_param")
            return ssd_ref.detection_output(boxes, prob, float(dp.get("confidence_threshold", 0.01)),
                                            float(nms.get("nms_threshold", 0.45)), int(nms.get("top_k", 400)),
                                            int(dp.get("keep_top_k", 200)))
        else:
            raise ValueError(f"oracle: layer type {typ}")
        for tp in top:
            t[tp] = y
    raise ValueError("oracle: no DetectionOutput layer")
