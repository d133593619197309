"""Weights for the classifier: seeded generator, checkpoint reader, and packer.

The reference loads ``weights/best_model.pth`` (``{'model_state_dict': ...}`` or a
bare state dict, keys prefixed ``net.``; reference deepfake_detection.py:44-51,
train.py:1034-1055).  That file is not in the reference tree, so benchmarks and
tests use `seeded_state_dict` - random-init weights of the same architecture and
key names.  `pack_b0` turns any such state dict into the kernel-ready blob the
C ABI (`dfd_create`, include/dfd_hip.h) takes: BatchNorm folded into the
preceding conv/linear in float64, weights re-laid out for NHWC kernels.

Blob layout (little endian): ``b"DFDW" u32 version u32 count`` then per tensor
``char name[48]; u32 ndim; u32 dims[4]; u64 offset; u64 nbytes`` and, 64-byte
aligned, the float32 payloads.
"""
from __future__ import annotations

import struct
from typing import Dict, Mapping

import numpy as np

from . import b0_arch as A

BLOB_MAGIC = b"DFDW"
BLOB_VERSION = 1
_NAME_LEN = 48


# --------------------------------------------------------------------------- generator
def seeded_state_dict(seed: int = 0) -> Dict[str, np.ndarray]:
    """Random-init state dict with the reference's key names (``net.*``).

    conv ~ N(0, g^2/fan_in) with a gain that keeps swish activations O(1),
    BN gamma~U(.8,1.2), beta~N(0,.1), running_mean~N(0,.1), running_var~U(.8,1.2),
    Linear ~ U(+-3/sqrt(fan_in)).  Deterministic in `seed`
    (numpy RandomState, independent of the torch version).
    """
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = {}

    def conv(name, co, ci_g, k, gain=1.6):
        fan_in = ci_g * k * k
        sd[name + ".weight"] = (rs.randn(co, ci_g, k, k) * (gain / np.sqrt(fan_in))).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = rs.uniform(0.8, 1.2, c).astype(np.float32)
        sd[name + ".bias"] = (rs.randn(c) * 0.1).astype(np.float32)
        sd[name + ".running_mean"] = (rs.randn(c) * 0.1).astype(np.float32)
        sd[name + ".running_var"] = rs.uniform(0.8, 1.2, c).astype(np.float32)
        sd[name + ".num_batches_tracked"] = np.array(0, dtype=np.int64)

    def bias(name, c, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        sd[name + ".bias"] = rs.uniform(-b, b, c).astype(np.float32)

    def linear(name, co, ci, gain=3.0):
        # torch's default U(+-1/sqrt(fan_in)) times a gain that spreads the logits over O(1)
        b = gain / np.sqrt(ci)
        sd[name + ".weight"] = rs.uniform(-b, b, (co, ci)).astype(np.float32)
        bias(name, co, ci)

    conv("net._conv_stem", A.STEM_OUT, 3, 3)
    bn("net._bn0", A.STEM_OUT)
    for b in A.BLOCKS:
        p = f"net._blocks.{b.index}"
        if b.expand != 1:
            conv(p + "._expand_conv", b.c_exp, b.c_in, 1)
            bn(p + "._bn0", b.c_exp)
        conv(p + "._depthwise_conv", b.c_exp, 1, b.kernel)
        bn(p + "._bn1", b.c_exp)
        conv(p + "._se_reduce", b.c_se, b.c_exp, 1, gain=1.0)
        bias(p + "._se_reduce", b.c_se, b.c_exp)
        conv(p + "._se_expand", b.c_exp, b.c_se, 1, gain=1.0)
        # a positive SE bias keeps the gate near 0.75 so the signal is not halved 16 times
        sd[p + "._se_expand.bias"] = (1.0 + rs.randn(b.c_exp) * 0.2).astype(np.float32)
        # project conv has no activation after it: unit gain; smaller on skip blocks
        conv(p + "._project_conv", b.c_out, b.c_exp, 1, gain=0.7 if b.skip else 1.0)
        bn(p + "._bn2", b.c_out)
    conv("net._conv_head", A.HEAD_OUT, A.BLOCKS[-1].c_out, 1)
    bn("net._bn1", A.HEAD_OUT)
    d = A.MLP_DIMS
    linear("net._fc.1", d[1], d[0])
    bn("net._fc.2", d[1])
    linear("net._fc.5", d[2], d[1])
    bn("net._fc.6", d[2])
    linear("net._fc.9", d[3], d[2])
    return sd


# --------------------------------------------------------------------------- checkpoint
def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
    """Read a reference-format checkpoint (reference deepfake_detection.py:44-51)."""
    import torch

    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    state = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    return {k: v.detach().cpu().numpy() for k, v in state.items()}


def to_torch(sd: Mapping[str, np.ndarray]):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


# --------------------------------------------------------------------------- packing
def _fold(w: np.ndarray, b, bnp: Mapping[str, np.ndarray], prefix: str, eps: float):
    """Fold eval-mode BatchNorm `prefix` into (w, b); w's axis 0 is the output channel."""
    g = bnp[prefix + ".weight"].astype(np.float64)
    beta = bnp[prefix + ".bias"].astype(np.float64)
    mu = bnp[prefix + ".running_mean"].astype(np.float64)
    var = bnp[prefix + ".running_var"].astype(np.float64)
    a = g / np.sqrt(var + eps)
    w64 = w.astype(np.float64) * a.reshape((-1,) + (1,) * (w.ndim - 1))
    b64 = (0.0 if b is None else b.astype(np.float64)) * a + (beta - mu * a)
    return w64.astype(np.float32), b64.astype(np.float32)


def pack_b0_tensors(sd: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Kernel-ready tensors (see csrc/b0_plan.cpp for how each one is consumed)."""
    sd = {(k if k.startswith("net.") else "net." + k): np.asarray(v) for k, v in sd.items()}
    t: Dict[str, np.ndarray] = {}
    eps = A.BN_EPS_BACKBONE
    w, b = _fold(sd["net._conv_stem.weight"], None, sd, "net._bn0", eps)
    # stem: [co][ci][ky][kx] -> [ky][kx][ci][co]
    t["stem.w"] = np.ascontiguousarray(w.transpose(2, 3, 1, 0))
    t["stem.b"] = b
    for blk in A.BLOCKS:
        p = f"net._blocks.{blk.index}"
        q = f"b{blk.index}"
        if blk.expand != 1:
            w, b = _fold(sd[p + "._expand_conv.weight"], None, sd, p + "._bn0", eps)
            t[q + ".exp.w"] = np.ascontiguousarray(w.reshape(blk.c_exp, blk.c_in))     # [co][ci]
            t[q + ".exp.b"] = b
        w, b = _fold(sd[p + "._depthwise_conv.weight"], None, sd, p + "._bn1", eps)
        # depthwise: [c][1][ky][kx] -> [ky][kx][c]
        t[q + ".dw.w"] = np.ascontiguousarray(w.reshape(blk.c_exp, blk.kernel, blk.kernel).transpose(1, 2, 0))
        t[q + ".dw.b"] = b
        t[q + ".se.w1"] = np.ascontiguousarray(sd[p + "._se_reduce.weight"].reshape(blk.c_se, blk.c_exp).astype(np.float32))
        t[q + ".se.b1"] = sd[p + "._se_reduce.bias"].astype(np.float32)
        # second SE matmul stored [c_se][c_exp] so consecutive lanes read consecutive channels
        t[q + ".se.w2"] = np.ascontiguousarray(sd[p + "._se_expand.weight"].reshape(blk.c_exp, blk.c_se).T.astype(np.float32))
        t[q + ".se.b2"] = sd[p + "._se_expand.bias"].astype(np.float32)
        w, b = _fold(sd[p + "._project_conv.weight"], None, sd, p + "._bn2", eps)
        t[q + ".proj.w"] = np.ascontiguousarray(w.reshape(blk.c_out, blk.c_exp))
        t[q + ".proj.b"] = b
    w, b = _fold(sd["net._conv_head.weight"], None, sd, "net._bn1", eps)
    t["head.w"] = np.ascontiguousarray(w.reshape(A.HEAD_OUT, A.BLOCKS[-1].c_out))
    t["head.b"] = b
    w, b = _fold(sd["net._fc.1.weight"], sd["net._fc.1.bias"], sd, "net._fc.2", A.BN_EPS_HEAD)
    t["fc1.w"], t["fc1.b"] = w, b
    w, b = _fold(sd["net._fc.5.weight"], sd["net._fc.5.bias"], sd, "net._fc.6", A.BN_EPS_HEAD)
    t["fc2.w"], t["fc2.b"] = w, b
    t["fc3.w"] = sd["net._fc.9.weight"].astype(np.float32)
    t["fc3.b"] = sd["net._fc.9.bias"].astype(np.float32)
    return t


def serialize(tensors: Mapping[str, np.ndarray]) -> bytes:
    names = list(tensors)
    entry = struct.Struct(f"<{_NAME_LEN}sI4IQQ")
    head = 12 + entry.size * len(names)
    off = (head + 63) // 64 * 64
    table = []
    chunks = []
    for n in names:
        a = np.ascontiguousarray(tensors[n], dtype=np.float32)
        if a.ndim > 4 or len(n.encode()) >= _NAME_LEN:
            raise ValueError(f"tensor {n}: unsupported rank or name length")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        table.append(entry.pack(n.encode(), a.ndim, *dims, off, a.nbytes))
        chunks.append((off, a.tobytes()))
        off = (off + a.nbytes + 63) // 64 * 64
    buf = bytearray(off)
    buf[0:12] = BLOB_MAGIC + struct.pack("<II", BLOB_VERSION, len(names))
    pos = 12
    for e in table:
        buf[pos:pos + entry.size] = e
        pos += entry.size
    for o, c in chunks:
        buf[o:o + len(c)] = c
    return bytes(buf)


# --------------------------------------------------------------------------- SSD detector
def seeded_ssd_state_dict(seed: int = 0, background_bias: float = 4.0) -> Dict[str, np.ndarray]:
    """Random-init detector weights in Caffe layout (``<layer>.weight`` [co][ci][k][k], ``.bias``).

    He-normal convolutions; the background-class bias of every confidence head is raised so that
    a random network marks only a fraction of a percent of the 8732 priors as faces, as a
    trained detector would (otherwise every prior fires and NMS is all that is exercised).
    """
    from . import ssd_arch as S

    rs = np.random.RandomState(seed + 1000)
    sd: Dict[str, np.ndarray] = {}
    for name, kind, a in S.LAYERS:
        if kind == "conv":
            _, ci, co, k = a[:4]
            gain = np.sqrt(2.0) if a[7] else 1.0
            if a[8] is not None or name.endswith("p"):
                gain *= 0.7                                    # two branches are summed
            if name == "conv1":
                gain /= 50.0                                   # mean-subtracted pixels are O(50): bring activations to O(1)
            sd[name + ".weight"] = (rs.randn(co, ci, k, k) * gain / np.sqrt(ci * k * k)).astype(np.float32)
            sd[name + ".bias"] = (rs.randn(co) * 0.05).astype(np.float32)
        elif kind == "l2norm":
            sd[name + ".scale"] = np.full(a[1], S.NORM_SCALE_INIT, np.float32)
    for t, c, m, _, _, ars, _ in S.SOURCES:
        p = S.priors_per_cell(ars)
        sd[t + "_loc.weight"] = (rs.randn(p * 4, c, 3, 3) * 0.5 / np.sqrt(c * 9)).astype(np.float32)
        sd[t + "_loc.bias"] = (rs.randn(p * 4) * 0.2).astype(np.float32)
        sd[t + "_conf.weight"] = (rs.randn(p * 2, c, 3, 3) * 0.6 / np.sqrt(c * 9)).astype(np.float32)
        b = rs.randn(p * 2).astype(np.float32) * 0.1
        b[0::2] += background_bias                            # channel order per prior: (background, face)
        sd[t + "_conf.bias"] = b
    return sd


def pack_ssd_tensors(sd: Mapping[str, np.ndarray], arch=None) -> Dict[str, np.ndarray]:
    """Detector tensors for the GPU: GEMM convs as [co][ky][kx][ci], the first conv as [ky][kx][ci][co],
    per-source loc+conf heads fused into one [p*6][ky][kx][ci] matrix (loc rows first).  With an `arch` other than
    the built-in `ssd_arch` (caffe_io.build_arch: a deploy.prototxt) the layer table travels in the blob too
    ("ssd.plan" / "ssd.names" / "ssd.srcs" / "ssd.det", read by csrc/ssd_api.hip load_plan)."""
    from . import ssd_arch as S

    A = arch if arch is not None else S
    t: Dict[str, np.ndarray] = {}
    kinds = {"conv": 2, "maxpool": 1, "l2norm": 3, "affine": 4, "add": 5}
    index = {"data": -1}
    rows, first_conv = [], True
    for i, (name, kind, a) in enumerate(A.LAYERS):
        index[name] = i
        if len(name) > 31 or not name.isascii():
            raise ValueError(f"layer name {name!r} does not fit the 31-character plan entry")
        if kind == "conv":
            src, ci, co, k, st, pd, dl, relu, res = a
            w = np.asarray(sd[name + ".weight"], np.float32)
            if first_conv:                                   # the 3-channel convolution on the image
                t[f"ssd.{name}.w"] = np.ascontiguousarray(w.transpose(2, 3, 1, 0))
                rows.append((0, index[src], -1, ci, co, k, st, pd, dl, int(bool(relu))))
                first_conv = False
            else:
                t[f"ssd.{name}.w"] = np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(w.shape[0], -1)
                rows.append((2, index[src], index[res] if res is not None else -1, ci, co, k, st, pd, dl, int(bool(relu))))
            t[f"ssd.{name}.b"] = np.asarray(sd[name + ".bias"], np.float32)
        elif kind == "maxpool":
            src, k, st = a
            c = rows[index[src]][4]
            rows.append((1, index[src], -1, c, c, k, st, 0, 1, 0))
        elif kind == "l2norm":
            src, c = a
            t[f"ssd.{name}.scale"] = np.asarray(sd[name + ".scale"], np.float32)
            rows.append((3, index[src], -1, c, c, 1, 1, 0, 1, 0))
        elif kind == "affine":
            src, c, relu = a
            t[f"ssd.{name}.scale"] = np.asarray(sd[name + ".scale"], np.float32)
            t[f"ssd.{name}.shift"] = np.asarray(sd[name + ".shift"], np.float32)
            rows.append((4, index[src], -1, c, c, 1, 1, 0, 1, int(bool(relu))))
        elif kind == "add":
            src, other, c, relu = a
            rows.append((5, index[src], index[other], c, c, 1, 1, 0, 1, int(bool(relu))))
        else:
            raise ValueError(f"unknown detector layer kind {kind!r}")
    for src, c, m, _, _, ars, _ in A.SOURCES:
        w = np.concatenate([sd[src + "_loc.weight"], sd[src + "_conf.weight"]], 0).astype(np.float32)
        t[f"ssd.{src}.head.w"] = np.ascontiguousarray(w.transpose(0, 2, 3, 1)).reshape(w.shape[0], -1)
        t[f"ssd.{src}.head.b"] = np.concatenate([sd[src + "_loc.bias"], sd[src + "_conf.bias"]]).astype(np.float32)
    if A is not S:
        if len(A.SOURCES) != 6:
            raise ValueError("the detector kernels are built for six source maps")
        t["ssd.plan"] = np.asarray(rows, np.float32)
        names = np.zeros((len(rows), 32), np.float32)
        for i, (name, _, _) in enumerate(A.LAYERS):
            names[i, :len(name)] = [ord(ch) for ch in name]
        t["ssd.names"] = names
        srcs = []
        for src, c, m, mn, mx, ars, step in A.SOURCES:
            ar = list(ars) + [0.0] * (2 - len(ars))
            srcs.append((index[src], c, m, mn, mx, len(ars), ar[0], ar[1], step))
        t["ssd.srcs"] = np.asarray(srcs, np.float32)
        t["ssd.det"] = np.asarray([A.INPUT, *A.IN_SCALE, *A.IN_SHIFT, *A.VARIANCES, A.NMS_THRESHOLD, A.TOP_K, A.KEEP_TOP_K,
                                   A.CONF_THRESHOLD], np.float32)
    return t


# MTCNN (facenet-pytorch models/mtcnn.py: PNet / RNet / ONet), torch layouts: conv [co][ci][k][k], dense [out][in]
MTCNN_CONVS = {
    "pnet": [("conv1", 10, 3, 3), ("conv2", 16, 10, 3), ("conv3", 32, 16, 3), ("conv4_1", 2, 32, 1), ("conv4_2", 4, 32, 1)],
    "rnet": [("conv1", 28, 3, 3), ("conv2", 48, 28, 3), ("conv3", 64, 48, 2)],
    "onet": [("conv1", 32, 3, 3), ("conv2", 64, 32, 3), ("conv3", 64, 64, 3), ("conv4", 128, 64, 2)],
}
# dense layers: (name, out, in, (W, H, C) of the conv map the first one flattens, or None)
MTCNN_DENSE = {
    "pnet": [],
    "rnet": [("dense4", 128, 576, (3, 3, 64)), ("dense5_1", 2, 128, None), ("dense5_2", 4, 128, None)],
    "onet": [("dense5", 256, 1152, (3, 3, 128)), ("dense6_1", 2, 256, None), ("dense6_2", 4, 256, None),
             ("dense6_3", 10, 256, None)],
}
MTCNN_PRELU = {"pnet": [("prelu1", 10), ("prelu2", 16), ("prelu3", 32)],
               "rnet": [("prelu1", 28), ("prelu2", 48), ("prelu3", 64), ("prelu4", 128)],
               "onet": [("prelu1", 32), ("prelu2", 64), ("prelu3", 64), ("prelu4", 128), ("prelu5", 256)]}


def seeded_mtcnn_state_dict(seed: int = 0, head_bias=None) -> Dict[str, np.ndarray]:
    """Random-init MTCNN weights under facenet-pytorch's state_dict names (``pnet.conv1.weight`` ...).

    The face-probability heads get a negative face-vs-background bias so that a random cascade lets a few
    percent of the P-Net cells and a fraction of the R-/O-Net candidates through, as trained networks do
    (with symmetric heads half of all cells fire and only NMS is exercised; with none the stage is inert).
    """
    rs = np.random.RandomState(seed + 2000)
    sd: Dict[str, np.ndarray] = {}
    # (mean, std) of the face-vs-background logit per stage.  The default lets ~20 % of the P-Net cells and nearly every
    # R-Net candidate through: a stress cascade for the parity tests (hundreds of windows per crop reach every stage).
    # MTCNN_SELECTIVE is the funnel of a trained cascade on a crop with one face (a fraction of a percent of the
    # P-Net cells, a minority of the R-Net candidates): what bench.py times next to the stress figure.
    if head_bias is None:
        head_bias = {"pnet.conv4_1": (-0.9, 1.0), "rnet.dense5_1": (2.4, 1.5), "onet.dense6_1": (0.8, 1.5)}
    for net in ("pnet", "rnet", "onet"):
        for name, co, ci, k in MTCNN_CONVS[net]:
            q = f"{net}.{name}"
            gain = 1.0 if q in head_bias or name == "conv4_2" else np.sqrt(2.0)
            if name == "conv1":
                gain *= 2.5                                      # (x - 127.5) / 128 has std ~0.35
            sd[q + ".weight"] = (rs.randn(co, ci, k, k) * gain / np.sqrt(ci * k * k)).astype(np.float32)
            sd[q + ".bias"] = (rs.randn(co) * 0.05).astype(np.float32)
        for name, co, ci, _ in MTCNN_DENSE[net]:
            q = f"{net}.{name}"
            is_head = name.startswith("dense5_") or name.startswith("dense6_")
            gain = 1.0 if is_head else np.sqrt(2.0)
            sd[q + ".weight"] = (rs.randn(co, ci) * gain / np.sqrt(ci)).astype(np.float32)
            sd[q + ".bias"] = (rs.randn(co) * 0.05).astype(np.float32)
        for name, c in MTCNN_PRELU[net]:
            sd[f"{net}.{name}.weight"] = (0.25 + 0.05 * rs.randn(c)).astype(np.float32)
    _calibrate_mtcnn_heads(sd, rs, head_bias)
    for q in ("pnet.conv4_2", "rnet.dense5_2", "onet.dense6_2"):  # box regression: small offsets
        sd[q + ".weight"] *= 0.15
    return sd


MTCNN_SELECTIVE = {"pnet.conv4_1": (-2.4, 1.0), "rnet.dense5_1": (0.6, 1.5), "onet.dense6_1": (1.2, 1.5)}   # ~10 P-Net candidates per crop, 3 of 4 bench crops keep a face


def load_mtcnn_checkpoints(directory: str) -> Dict[str, np.ndarray]:
    """facenet-pytorch's ``data/pnet.pt``, ``rnet.pt``, ``onet.pt`` (plain state_dicts) -> one dict with the
    ``pnet.`` / ``rnet.`` / ``onet.`` prefixes `pack_mtcnn_tensors` expects."""
    import os

    import torch

    out: Dict[str, np.ndarray] = {}
    for net in ("pnet", "rnet", "onet"):
        sd = torch.load(os.path.join(directory, net + ".pt"), map_location="cpu")
        for k, v in sd.items():
            out[f"{net}.{k}"] = v.detach().cpu().numpy().astype(np.float32)
    return out


def _calibrate_mtcnn_heads(sd: Dict[str, np.ndarray], rs, targets) -> None:
    """Rescale / re-bias the three probability heads of a random cascade so that the face-vs-background logit
    has the requested mean and spread on noise inputs.  Synthetic-weight generation only: a trained
    checkpoint is used as it is.  (The positive mean of PReLU activations pushes the raw random heads by
    5-20 logits one way: every candidate passes, or none.)"""
    import torch
    import torch.nn.functional as F

    t = {k: torch.from_numpy(v) for k, v in sd.items()}

    def trunk(net, x):
        pools = {"pnet": [(2, 2)], "rnet": [(3, 2), (3, 2)], "onet": [(3, 2), (3, 2), (2, 2)]}[net]
        convs = [c for c in MTCNN_CONVS[net] if not c[0].startswith("conv4_")]
        for i, (name, _, _, _) in enumerate(convs):
            x = F.prelu(F.conv2d(x, t[f"{net}.{name}.weight"], t[f"{net}.{name}.bias"]), t[f"{net}.prelu{i + 1}.weight"])
            if i < len(pools):
                x = F.max_pool2d(x, pools[i][0], pools[i][1], ceil_mode=True)
        if net == "pnet":
            return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
        d = MTCNN_DENSE[net][0][0]
        x = x.permute(0, 3, 2, 1).reshape(x.shape[0], -1)
        return F.prelu(F.linear(x, t[f"{net}.{d}.weight"], t[f"{net}.{d}.bias"]), t[f"{net}.prelu{len(convs) + 1}.weight"])

    with torch.no_grad():
        for q, (mean, std) in targets.items():
            net = q.split(".")[0]
            size = {"pnet": 40, "rnet": 24, "onet": 48}[net]
            # image-like inputs: a per-sample colour offset and a smooth ramp under the pixel noise
            base = rs.randn(64, 3, 1, 1) * 0.3 + np.linspace(-0.3, 0.3, size).reshape(1, 1, 1, size) * rs.randn(64, 1, 1, 1)
            x = base + rs.randn(64, 3, size, size) * 0.15
            x[32:] = (rs.randint(50, 200, (32, 3, size, size)) - 127.5) * 0.0078125      # and plain pixel noise (bench frames)
            x = torch.from_numpy(x.astype(np.float32))
            feat = trunk(net, x)
            w = t[q + ".weight"].reshape(2, -1)
            d = feat @ (w[1] - w[0])
            k = std / float(d.std())
            sd[q + ".weight"] = (sd[q + ".weight"] * k).astype(np.float32)
            sd[q + ".bias"][0] = 0.0
            sd[q + ".bias"][1] = np.float32(mean - k * float(d.mean()))


def pack_mtcnn_tensors(sd: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Device layouts (names ``mtcnn.<net>.<layer>.w/.b/.a``): conv weights [ci][ky][kx][co] (consecutive output
    channels contiguous), dense weights [in][out] with `in` re-ordered from the package's (W, H, C) flatten
    (``x.permute(0, 3, 2, 1)``) to this build's NHWC (H, W, C) order, PReLU slopes as they are."""
    t: Dict[str, np.ndarray] = {}
    for net in ("pnet", "rnet", "onet"):
        for name, co, ci, k in MTCNN_CONVS[net]:
            w = np.asarray(sd[f"{net}.{name}.weight"], np.float32)
            if w.shape != (co, ci, k, k):
                raise ValueError(f"{net}.{name}.weight: shape {w.shape}")
            t[f"mtcnn.{net}.{name}.w"] = np.ascontiguousarray(w.transpose(1, 2, 3, 0))
            t[f"mtcnn.{net}.{name}.b"] = np.asarray(sd[f"{net}.{name}.bias"], np.float32)
        for name, co, ci, whc in MTCNN_DENSE[net]:
            w = np.asarray(sd[f"{net}.{name}.weight"], np.float32)
            if w.shape != (co, ci):
                raise ValueError(f"{net}.{name}.weight: shape {w.shape}")
            if whc is not None:
                W_, H_, C_ = whc
                w = w.reshape(co, W_, H_, C_).transpose(0, 2, 1, 3).reshape(co, ci)      # (w,h,c) -> (h,w,c)
            t[f"mtcnn.{net}.{name}.w"] = np.ascontiguousarray(w.T)
            t[f"mtcnn.{net}.{name}.b"] = np.asarray(sd[f"{net}.{name}.bias"], np.float32)
        for name, c in MTCNN_PRELU[net]:
            t[f"mtcnn.{net}.{name}.a"] = np.asarray(sd[f"{net}.{name}.weight"], np.float32).reshape(c)

    # The R-/O-Net convolutions behind the first one run on the MFMA implicit-GEMM kernel (C_in % 32 == 0):
    # weights [co][ky][kx][ci] with channels zero-padded to 32 / 64 (R-Net: 28 -> 32, 48 -> 64; a padded channel
    # is bias 0 -> PReLU(0) = 0 and meets zero weights downstream).
    def pad_to(a, n, axis, fill=0.0):
        padw = [(0, 0)] * a.ndim
        padw[axis] = (0, n - a.shape[axis])
        return np.pad(a, padw, constant_values=fill)

    def gemm_conv(net, name, prelu, co_pad, ci_pad):
        w = np.asarray(sd[f"{net}.{name}.weight"], np.float32).transpose(0, 2, 3, 1)          # [co][ky][kx][ci]
        t[f"mtcnn.{net}.{name}.wg"] = np.ascontiguousarray(pad_to(pad_to(w, ci_pad, 3), co_pad, 0))
        t[f"mtcnn.{net}.{name}.bg"] = pad_to(np.asarray(sd[f"{net}.{name}.bias"], np.float32), co_pad, 0)
        t[f"mtcnn.{net}.{name}.ag"] = pad_to(np.asarray(sd[f"{net}.{prelu}.weight"], np.float32).reshape(-1), co_pad, 0)

    # P-Net conv2 / conv3 for the MFMA kernel (mt_pnet_mfma_kernel): [co][ky * KR + kx * ci_n + ci], a kernel row's
    # 3 * ci_n values contiguous (as they are in the NHWC map), rows padded to KR = 32 / 48, K to a multiple of 32
    def pnet_mfma(name, co, ci_n, kr):
        w = np.asarray(sd[f"pnet.{name}.weight"], np.float32)                                 # [co][ci][ky][kx]
        rows = w.transpose(0, 2, 3, 1).reshape(co, 3, 3 * ci_n)                               # [co][ky][kx * ci_n + ci]
        rows = pad_to(rows, kr, 2).reshape(co, 3 * kr)
        t[f"mtcnn.pnet.{name}.wm"] = np.ascontiguousarray(pad_to(rows, (3 * kr + 31) // 32 * 32, 1))

    pnet_mfma("conv2", 16, 10, 32)
    pnet_mfma("conv3", 32, 16, 48)
    w1 = t["mtcnn.rnet.conv1.w"]                                                              # [ci][ky][kx][28] -> 32
    t["mtcnn.rnet.conv1.wp"] = np.ascontiguousarray(pad_to(w1, 32, 3))
    t["mtcnn.rnet.conv1.bp"] = pad_to(t["mtcnn.rnet.conv1.b"], 32, 0)
    t["mtcnn.rnet.conv1.ap"] = pad_to(t["mtcnn.rnet.prelu1.a"], 32, 0)
    gemm_conv("rnet", "conv2", "prelu2", 64, 32)
    gemm_conv("rnet", "conv3", "prelu3", 64, 64)
    gemm_conv("onet", "conv2", "prelu2", 64, 32)
    gemm_conv("onet", "conv3", "prelu3", 64, 64)
    gemm_conv("onet", "conv4", "prelu4", 128, 64)
    return t


def pack_all(b0_sd: Mapping[str, np.ndarray], ssd_sd: Mapping[str, np.ndarray] = None,
             mtcnn_sd: Mapping[str, np.ndarray] = None, ssd_arch=None, haar: Mapping[str, np.ndarray] = None) -> bytes:
    """One blob for `dfd_create`: classifier + colour tables (+ detector, + MTCNN cascade, + Haar cascade when given).
    `ssd_arch`: the detector topology `ssd_sd` belongs to (default: the built-in `ssd_arch` module);
    `haar`: the arrays of `haar.load_cascade_xml` (the reference's fallback detector)."""
    from . import luts

    t = pack_b0_tensors(b0_sd)
    t.update(luts.as_float_tensors())
    if ssd_sd is not None:
        t.update(pack_ssd_tensors(ssd_sd, ssd_arch))
    if mtcnn_sd is not None:
        t.update(pack_mtcnn_tensors(mtcnn_sd))
    if haar is not None:
        t.update({k: np.asarray(v, np.float32) for k, v in haar.items()})
    return serialize(t)


def pack_b0(sd: Mapping[str, np.ndarray], with_tables: bool = True) -> bytes:
    """Classifier blob; by default also carries the colour LUTs the pre-processing kernels use."""
    t = pack_b0_tensors(sd)
    if with_tables:
        from . import luts

        t.update(luts.as_float_tensors())
    return serialize(t)
