"""Caffe SSD face detector -> this build's detector plan.

The reference loads OpenCV's res10 SSD from two files (reference face_detection.py:19-24):
``weights/deploy.prototxt`` + ``weights/res10_300x300_ssd_iter_140000_fp16.caffemodel`` through
``cv2.dnn.readNetFromCaffe``.  Neither cv2 nor protobuf definitions for Caffe are dependencies here, so this
module reads both formats itself:

  * `parse_prototxt`   protobuf TEXT format -> nested `Msg` (generic, schema-free);
  * `parse_caffemodel` protobuf WIRE format of `NetParameter`: layer name -> list of blobs (float32), from
                       `layer` (field 100) or V1 `layers` (field 2); blob payloads as packed / unpacked `data`,
                       `double_data`, or OpenCV's fp16 extension (`raw_data_type` = FLOAT16, `raw_data`);
  * `build_arch`       the layer graph of the prototxt -> (arch, state_dict) in the vocabulary of `ssd_arch`:
                       BatchNorm+Scale pairs folded into the convolution before them where that convolution has no
                       other reader, kept as "affine" layers otherwise (pre-activation ResNet blocks), in-place
                       ReLUs attached to their producer, Eltwise SUM fused into one operand's convolution as its
                       residual, Normalize -> "l2norm", the mbox branch (Permute/Flatten/Concat/Reshape/Softmax)
                       recognised structurally from the DetectionOutput layer backwards, PriorBox and
                       DetectionOutput parameters read from their layers.

`weights.pack_all(b0_sd, ssd_sd, mtcnn_sd, ssd_arch=arch)` then ships the plan inside the weights blob and
csrc/ssd_api.hip executes it (constraints of the kernels - 300x300 input, 7x7/2 first convolution with 32 outputs,
3x3/2 max pooling, C_in % 32 == 0, 3x3 pad-1 heads, six sources, top_k 400 - are checked here and again in C).
The topology is whatever the prototxt says: nothing about res10 is hard-coded.
"""
from __future__ import annotations

import math
import struct
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------ text format
class Msg:
    """A protobuf text message: ordered (key, value) pairs; values are str / float / int / bool / Msg."""

    def __init__(self):
        self.items: List[Tuple[str, object]] = []

    def getall(self, key):
        return [v for k, v in self.items if k == key]

    def get(self, key, default=None):
        for k, v in self.items:
            if k == key:
                return v
        return default

    def __contains__(self, key):
        return any(k == key for k, _ in self.items)


def _tokens(text: str):
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if c.isspace() or c in ",;":
            i += 1
        elif c == "#":
            while i < n and text[i] != "\n":
                i += 1
        elif c in "{}:<>":
            yield c
            i += 1
        elif c in "\"'":
            j = i + 1
            out = []
            while j < n and text[j] != c:
                if text[j] == "\\" and j + 1 < n:
                    j += 1
                out.append(text[j])
                j += 1
            yield ("str", "".join(out))
            i = j + 1
        else:
            j = i
            while j < n and not text[j].isspace() and text[j] not in "{}:<>#,;\"'":
                j += 1
            yield text[i:j]
            i = j


def _scalar(tok):
    if isinstance(tok, tuple):
        return tok[1]
    if tok in ("true", "True"):
        return True
    if tok in ("false", "False"):
        return False
    try:
        return int(tok)
    except ValueError:
        try:
            return float(tok)
        except ValueError:
            return tok                      # enum identifier


def parse_prototxt(text: str) -> Msg:
    toks = list(_tokens(text))
    pos = 0

    def message(closer):
        nonlocal pos
        m = Msg()
        while pos < len(toks):
            t = toks[pos]
            if t == closer:
                pos += 1
                return m
            if not isinstance(t, str) or t in "{}:<>":
                raise ValueError(f"prototxt: unexpected token {t!r}")
            key = t
            pos += 1
            if pos < len(toks) and toks[pos] == ":":
                pos += 1
            if pos >= len(toks):
                raise ValueError("prototxt: truncated")
            if toks[pos] in ("{", "<"):
                close = "}" if toks[pos] == "{" else ">"
                pos += 1
                m.items.append((key, message(close)))
            else:
                m.items.append((key, _scalar(toks[pos])))
                pos += 1
        if closer is not None:
            raise ValueError("prototxt: missing closing brace")
        return m

    return message(None)


# ------------------------------------------------------------------------------------------ wire format
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("caffemodel: bad varint")


def _fields(buf):
    """yields (field number, wire type, value) - value: int (varint / fixed) or memoryview (length-delimited)"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise ValueError("caffemodel: truncated field")
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        else:
            raise ValueError(f"caffemodel: unsupported wire type {wt}")
        yield num, wt, v


def _blob(buf) -> np.ndarray:
    shape, legacy = None, {}
    data, ddata, raw, raw_type = [], [], None, None
    for num, wt, v in _fields(buf):
        if num == 7 and wt == 2:                                   # BlobShape
            dims = []
            for n2, w2, v2 in _fields(v):
                if n2 == 1:
                    if w2 == 2:
                        p = 0
                        while p < len(v2):
                            d, p = _varint(v2, p)
                            dims.append(d)
                    else:
                        dims.append(v2)
            shape = dims
        elif num == 5:
            data.append(np.frombuffer(v, "<f4") if wt == 2 else np.frombuffer(v, "<f4", 1))
        elif num == 8:
            ddata.append(np.frombuffer(v, "<f8") if wt == 2 else np.frombuffer(v, "<f8", 1))
        elif num == 10 and wt == 0:
            raw_type = v
        elif num == 12 and wt == 2:
            raw = bytes(v)
        elif num in (1, 2, 3, 4) and wt == 0:
            legacy[num] = v
    if raw is not None:
        dt = {0: "<f8", 1: "<f4", 2: "<f2", 3: "<i4", 4: "<u4"}.get(raw_type if raw_type is not None else 1)
        if dt is None:
            raise ValueError(f"caffemodel: raw_data_type {raw_type}")
        arr = np.frombuffer(raw, dt).astype(np.float32)
    elif data:
        arr = np.concatenate(data).astype(np.float32)
    elif ddata:
        arr = np.concatenate(ddata).astype(np.float32)
    else:
        arr = np.zeros(0, np.float32)
    if shape is None and legacy:
        shape = [legacy.get(k, 1) for k in (1, 2, 3, 4)]
    if shape is not None and int(np.prod(shape)) == arr.size:
        arr = arr.reshape(shape)
    return arr


def parse_caffemodel(data: bytes) -> Dict[str, List[np.ndarray]]:
    """layer name -> its blobs, in file order"""
    out: Dict[str, List[np.ndarray]] = {}
    for num, wt, v in _fields(memoryview(data)):
        if wt != 2 or num not in (100, 2):
            continue
        name_f, blob_f = (1, 7) if num == 100 else (4, 6)
        name, blobs = None, []
        for n2, w2, v2 in _fields(v):
            if n2 == name_f and w2 == 2:
                name = bytes(v2).decode("utf-8", "replace")
            elif n2 == blob_f and w2 == 2:
                blobs.append(_blob(v2))
        if name is not None and blobs:
            out[name] = blobs
    return out


# ------------------------------------------------------------------------------------------ graph -> plan
class _Node:
    def __init__(self, name, op, inputs, **kw):
        self.name, self.op, self.inputs = name, op, list(inputs)
        self.relu = False
        self.res: Optional["_Node"] = None
        self.__dict__.update(kw)


def _rep(msg: Msg, key, default):
    v = msg.getall(key)
    return v if v else [default]


def build_arch(prototxt: str, blobs: Dict[str, List[np.ndarray]], mean_bgr=(104.0, 177.0, 123.0)):
    """-> (arch, state_dict).  `arch` has the attributes of the `ssd_arch` module (INPUT, LAYERS, SOURCES, ...)."""
    net = parse_prototxt(prototxt)
    layers = net.getall("layer")
    if not layers:
        raise ValueError("prototxt: no `layer` entries (V1 `layers` prototxts are not supported)")
    # ---- input
    in_name, in_dims = None, None
    if "input" in net:
        in_name = net.get("input")
        shp = net.get("input_shape")
        in_dims = [int(d) for d in shp.getall("dim")] if shp is not None else [int(d) for d in net.getall("input_dim")]
    prod: Dict[str, _Node] = {}                 # blob name -> node currently producing it
    nodes: List[_Node] = []
    det = None
    meta = []                                   # shape-only ops of the mbox branch, for the walk back

    def src(blob):
        if blob not in prod:
            raise ValueError(f"prototxt: blob {blob!r} is read before any layer writes it")
        return prod[blob]

    for L in layers:
        name, typ = L.get("name"), L.get("type")
        bottoms, tops = L.getall("bottom"), L.getall("top")
        if typ == "Input":
            in_name = tops[0]
            in_dims = [int(d) for d in L.get("input_param").get("shape").getall("dim")]
            continue
        if in_name is not None and in_name not in prod:
            prod[in_name] = _Node("data", "data", [])
        if typ == "Convolution":
            p = L.get("convolution_param")
            k = int(_rep(p, "kernel_size", p.get("kernel_h", 1))[0])
            node = _Node(name, "conv", [src(bottoms[0])], co=int(p.get("num_output")), k=k,
                         stride=int(_rep(p, "stride", 1)[0]), pad=int(_rep(p, "pad", 0)[0]),
                         dil=int(_rep(p, "dilation", 1)[0]), bias_term=bool(p.get("bias_term", True)),
                         group=int(p.get("group", 1)))
            if node.group != 1:
                raise ValueError(f"{name}: grouped convolutions are not supported")
            w = blobs[name][0]
            node.W = w.reshape(node.co, -1, k, k).astype(np.float32)
            node.b = blobs[name][1].reshape(-1).astype(np.float32) if node.bias_term and len(blobs[name]) > 1 else np.zeros(node.co, np.float32)
        elif typ == "BatchNorm":
            m, v = blobs[name][0].reshape(-1).astype(np.float64), blobs[name][1].reshape(-1).astype(np.float64)
            sf = float(blobs[name][2].reshape(-1)[0]) if len(blobs[name]) > 2 else 1.0
            sf = 1.0 / sf if sf != 0 else 0.0
            eps = float(L.get("batch_norm_param").get("eps", 1e-5)) if "batch_norm_param" in L else 1e-5
            a = 1.0 / np.sqrt(v * sf + eps)
            node = _Node(name, "affine", [src(bottoms[0])], scale=a, shift=-(m * sf) * a)
        elif typ == "Scale":
            g = blobs[name][0].reshape(-1).astype(np.float64)
            has_b = bool(L.get("scale_param").get("bias_term", False)) if "scale_param" in L else False
            b = blobs[name][1].reshape(-1).astype(np.float64) if has_b and len(blobs[name]) > 1 else np.zeros_like(g)
            node = _Node(name, "affine", [src(bottoms[0])], scale=g, shift=b)
        elif typ == "ReLU":
            if "relu_param" in L and float(L.get("relu_param").get("negative_slope", 0)) != 0:
                raise ValueError(f"{name}: leaky ReLU is not supported")
            node = _Node(name, "relu", [src(bottoms[0])])
        elif typ == "Pooling":
            p = L.get("pooling_param")
            if p.get("pool", "MAX") not in ("MAX", 0):
                raise ValueError(f"{name}: only MAX pooling is supported")
            node = _Node(name, "pool", [src(bottoms[0])], k=int(p.get("kernel_size")), stride=int(p.get("stride", 1)), pad=int(p.get("pad", 0)))
        elif typ == "Eltwise":
            op = L.get("eltwise_param").get("operation", "SUM") if "eltwise_param" in L else "SUM"
            if op not in ("SUM", 1) or len(bottoms) != 2:
                raise ValueError(f"{name}: only a two-operand Eltwise SUM is supported")
            node = _Node(name, "add", [src(bottoms[0]), src(bottoms[1])])
        elif typ == "Normalize":
            p = L.get("norm_param")
            if p is not None and (bool(p.get("across_spatial", True)) or bool(p.get("channel_shared", True))):
                raise ValueError(f"{name}: Normalize must be per pixel with per-channel scales")
            node = _Node(name, "l2norm", [src(bottoms[0])], scale=blobs[name][0].reshape(-1).astype(np.float32))
        elif typ in ("Permute", "Flatten", "Reshape", "Softmax", "Concat"):
            node = _Node(name, typ.lower(), [src(b) for b in bottoms], msg=L)
            meta.append(node)
        elif typ == "PriorBox":
            node = _Node(name, "priorbox", [src(bottoms[0])], msg=L.get("prior_box_param"))
            meta.append(node)
        elif typ == "DetectionOutput":
            node = _Node(name, "detout", [src(b) for b in bottoms], msg=L.get("detection_output_param"))
            det = node
        else:
            raise ValueError(f"layer {name}: type {typ} is not supported by the detector plan")
        nodes.append(node)
        for t in tops:
            prod[t] = node
    if det is None:
        raise ValueError("prototxt: no DetectionOutput layer")
    if not in_dims or len(in_dims) != 4 or in_dims[1] != 3 or in_dims[2] != in_dims[3]:
        raise ValueError(f"prototxt: input shape {in_dims} (a square 3-channel image is expected)")

    compute = [n for n in nodes if n.op in ("conv", "affine", "relu", "pool", "add", "l2norm")]

    def consumers(n):
        return [m for m in nodes if n in m.inputs]

    def replace(old, new):
        for m in nodes:
            m.inputs = [new if i is old else i for i in m.inputs]
            if m.res is old:
                m.res = new

    # ---- 1. affine o affine, affine into the convolution before it, affine on the data blob into the input transform
    in_scale, in_shift = np.ones(3), -np.asarray(mean_bgr, np.float64)
    changed = True
    while changed:
        changed = False
        for n in list(compute):
            if n.op != "affine":
                continue
            p = n.inputs[0]
            if p.op == "data" and len(consumers(p)) == 1:
                in_shift = in_shift * n.scale + n.shift
                in_scale = in_scale * n.scale
            elif p.op == "affine" and len(consumers(p)) == 1 and not p.relu:
                p.shift = p.shift * n.scale + n.shift
                p.scale = p.scale * n.scale
            elif p.op == "conv" and len(consumers(p)) == 1 and not p.relu and p.res is None:
                p.W = (p.W.astype(np.float64) * n.scale[:, None, None, None]).astype(np.float32)
                p.b = (p.b.astype(np.float64) * n.scale + n.shift).astype(np.float32)
            else:
                continue
            replace(n, p)
            compute.remove(n)
            nodes.remove(n)
            changed = True
    # ---- 2. Eltwise SUM into one operand's convolution (its residual), 3. ReLU onto its producer
    changed = True
    while changed:
        changed = False
        for n in list(compute):
            if n.op == "add":
                for a, b in ((n.inputs[0], n.inputs[1]), (n.inputs[1], n.inputs[0])):
                    if a.op == "conv" and len(consumers(a)) == 1 and not a.relu and a.res is None and a.inputs[0].op != "data":
                        a.res = b
                        replace(n, a)
                        compute.remove(n)
                        nodes.remove(n)
                        changed = True
                        break
            elif n.op == "relu":
                p = n.inputs[0]
                if p.op in ("conv", "affine", "add") and len(consumers(p)) == 1 and not p.relu:
                    p.relu = True
                    replace(n, p)
                    compute.remove(n)
                    nodes.remove(n)
                    changed = True
                else:
                    raise ValueError(f"ReLU {n.name}: its input {p.name} ({p.op}) has other readers or cannot carry an activation")
    # ---- heads: DetectionOutput <- (loc concat, conf ... concat, priorbox concat)
    def back(n, through):
        while n.op in through:
            n = n.inputs[0]
        return n

    loc_cat = back(det.inputs[0], ("flatten", "permute", "reshape"))
    conf_cat = back(det.inputs[1], ("flatten", "softmax", "reshape", "permute"))
    prior_cat = back(det.inputs[2], ("flatten", "reshape"))
    if not (loc_cat.op == conf_cat.op == prior_cat.op == "concat") or not (len(loc_cat.inputs) == len(conf_cat.inputs) == len(prior_cat.inputs)):
        raise ValueError("prototxt: the mbox branch does not have the SSD loc / conf / priorbox concat structure")
    head_nodes, sources_raw = set(), []
    for lo, co, pr in zip(loc_cat.inputs, conf_cat.inputs, prior_cat.inputs):
        lc, cc = back(lo, ("flatten", "permute")), back(co, ("flatten", "permute"))
        if lc.op != "conv" or cc.op != "conv" or pr.op != "priorbox":
            raise ValueError("prototxt: a detection head is not Convolution -> Permute -> Flatten")
        s = lc.inputs[0]
        if cc.inputs[0] is not s or pr.inputs[0] is not s:
            raise ValueError(f"prototxt: loc / conf / priorbox of head {lc.name} read different tensors")
        for c in (lc, cc):
            if not (c.k == 3 and c.pad == 1 and c.stride == 1 and c.dil == 1) or c.relu or c.res is not None:
                raise ValueError(f"head {c.name}: must be a plain 3x3 pad-1 convolution")
            head_nodes.add(c)
        sources_raw.append((s, lc, cc, pr.msg))
    # ---- emit in dependency order
    order: List[_Node] = []
    seen = set()

    def emit(n):
        if id(n) in seen or n.op == "data":
            return
        seen.add(id(n))
        for i in n.inputs:
            emit(i)
        if n.res is not None:
            emit(n.res)
        order.append(n)

    for n in compute:
        if n not in head_nodes:
            emit(n)
    chans: Dict[int, int] = {}
    LAYERS, sd = [], {}
    tname = lambda n: "data" if n.op == "data" else n.name                    # noqa: E731
    for n in order:
        cin = 3 if n.inputs[0].op == "data" else chans[id(n.inputs[0])]
        if n.op == "conv":
            if n.W.shape[1] != cin:
                raise ValueError(f"{n.name}: weights have {n.W.shape[1]} input channels, its input has {cin}")
            LAYERS.append((n.name, "conv", (tname(n.inputs[0]), cin, n.co, n.k, n.stride, n.pad, n.dil, n.relu, tname(n.res) if n.res is not None else None)))
            sd[n.name + ".weight"], sd[n.name + ".bias"] = n.W, n.b
            chans[id(n)] = n.co
        elif n.op == "pool":
            if not (n.k == 3 and n.stride == 2 and n.pad == 0):
                raise ValueError(f"{n.name}: pooling must be 3x3 stride 2 pad 0")
            LAYERS.append((n.name, "maxpool", (tname(n.inputs[0]), n.k, n.stride)))
            chans[id(n)] = cin
        elif n.op == "affine":
            LAYERS.append((n.name, "affine", (tname(n.inputs[0]), cin, n.relu)))
            sd[n.name + ".scale"], sd[n.name + ".shift"] = n.scale.astype(np.float32), n.shift.astype(np.float32)
            chans[id(n)] = cin
        elif n.op == "add":
            LAYERS.append((n.name, "add", (tname(n.inputs[0]), tname(n.inputs[1]), cin, n.relu)))
            chans[id(n)] = cin
        elif n.op == "l2norm":
            LAYERS.append((n.name, "l2norm", (tname(n.inputs[0]), cin)))
            sd[n.name + ".scale"] = n.scale
            chans[id(n)] = cin
    # ---- sources (needs the spatial sizes: run the shape arithmetic of ssd_arch)
    size = {"data": in_dims[2]}
    for name, kind, a in LAYERS:
        if kind == "conv":
            size[name] = (size[a[0]] + 2 * a[5] - a[6] * (a[3] - 1) - 1) // a[4] + 1
        elif kind == "maxpool":
            size[name] = -(-(size[a[0]] - a[1]) // a[2]) + 1
        else:
            size[name] = size[a[0]]
    SOURCES, variances = [], None
    for s, lc, cc, pm in sources_raw:
        ars = [float(v) for v in pm.getall("aspect_ratio")]
        ars = [a for a in ars if abs(a - 1.0) > 1e-6]
        if not bool(pm.get("flip", True)) or bool(pm.get("clip", False)) or abs(float(pm.get("offset", 0.5)) - 0.5) > 1e-6:
            raise ValueError("PriorBox: flip must be true, clip false, offset 0.5")
        mn, mx = float(pm.get("min_size")), float(pm.get("max_size"))
        p = 2 + 2 * len(ars)
        if lc.co != 4 * p or cc.co != 2 * p:
            raise ValueError(f"head of {s.name}: {lc.co}/{cc.co} outputs for {p} priors per cell and two classes")
        m = size[s.name]
        step = float(pm.get("step")) if "step" in pm else in_dims[2] / m
        v = [float(x) for x in pm.getall("variance")] or [0.1, 0.1, 0.2, 0.2]
        variances = v if len(v) == 4 else [v[0]] * 4
        SOURCES.append((s.name, chans[id(s)], m, mn, mx, tuple(ars), step))
        sd[s.name + "_loc.weight"], sd[s.name + "_loc.bias"] = lc.W, lc.b
        sd[s.name + "_conf.weight"], sd[s.name + "_conf.bias"] = cc.W, cc.b
    dp = det.msg
    nms = dp.get("nms_param")
    if int(dp.get("num_classes", 2)) != 2 or int(dp.get("background_label_id", 0)) != 0 or not bool(dp.get("share_location", True)):
        raise ValueError("DetectionOutput: two classes, background 0, shared location expected")
    if dp.get("code_type", "CORNER") not in ("CENTER_SIZE", 2):
        raise ValueError("DetectionOutput: code_type CENTER_SIZE expected")
    arch = SimpleNamespace(
        INPUT=in_dims[2], MEAN_BGR=tuple(float(m) for m in mean_bgr), IN_SCALE=tuple(float(v) for v in in_scale),
        IN_SHIFT=tuple(float(v) for v in in_shift), NUM_CLASSES=2, VARIANCES=tuple(variances),
        NMS_THRESHOLD=float(nms.get("nms_threshold", 0.45)) if nms is not None else 0.45,
        TOP_K=int(nms.get("top_k", 400)) if nms is not None else 400, KEEP_TOP_K=int(dp.get("keep_top_k", 200)),
        CONF_THRESHOLD=float(dp.get("confidence_threshold", 0.01)), LAYERS=LAYERS, SOURCES=SOURCES)
    arch.priors_per_cell = lambda ars: 2 + 2 * len(ars)
    return arch, sd


def load_caffe_detector(prototxt_path: str, caffemodel_path: str):
    """reference face_detection.py:19-24: (deploy.prototxt, *.caffemodel) -> (arch, state_dict)"""
    with open(prototxt_path) as f:
        text = f.read()
    with open(caffemodel_path, "rb") as f:
        blobs = parse_caffemodel(f.read())
    return build_arch(text, blobs)
