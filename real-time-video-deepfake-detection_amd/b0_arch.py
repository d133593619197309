"""EfficientNet-B0 block table for the deepfake classifier (host-side constants).

The reference builds its backbone with ``EfficientNet.from_name('efficientnet-b0')``
(reference model.py:39-43) and replaces ``_fc`` by a 1280->512->256->1 MLP
(reference model.py:50-61).  ``efficientnet_pytorch`` is not vendored in the
reference tree; the table below restates its published B0 block arguments
(r1_k3_s11_e1_i32_o16 ... r1_k3_s11_e6_i192_o320, se_ratio 0.25, image_size 224,
BN eps 1e-3) expanded to one row per block, with TF-"SAME" static padding
resolved for a 224x224 input.  The C++ side (csrc/b0_plan.cpp) carries the same
table; `tests/test_weights.py` checks that both agree and that the parameter
count is 4,796,541.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

IMAGE_SIZE = 224
STEM_OUT = 32
HEAD_OUT = 1280
BN_EPS_BACKBONE = 1e-3      # efficientnet_pytorch batch_norm_epsilon
BN_EPS_HEAD = 1e-5          # torch.nn.BatchNorm1d default (reference model.py:53,57)
SE_RATIO = 0.25
MLP_DIMS = (1280, 512, 256, 1)   # reference model.py:50-61

# (repeats, kernel, stride, expand, in, out)
_STAGES = (
    (1, 3, 1, 1, 32, 16),
    (2, 3, 2, 6, 16, 24),
    (2, 5, 2, 6, 24, 40),
    (3, 3, 2, 6, 40, 80),
    (3, 5, 1, 6, 80, 112),
    (4, 5, 2, 6, 112, 192),
    (1, 3, 1, 6, 192, 320),
)


@dataclass(frozen=True)
class Block:
    index: int
    kernel: int
    stride: int
    expand: int
    c_in: int
    c_out: int
    c_exp: int
    c_se: int
    h_in: int
    h_out: int
    pad_lo: int      # zero rows/cols added before the first input row/col
    pad_hi: int      # ... after the last one (TF SAME puts the odd one here)
    skip: bool


def same_pad(size_in: int, kernel: int, stride: int):
    """TF-"SAME" padding as efficientnet_pytorch's Conv2dStaticSamePadding does it."""
    out = -(-size_in // stride)
    total = max((out - 1) * stride + kernel - size_in, 0)
    return out, total // 2, total - total // 2


def blocks() -> List[Block]:
    out: List[Block] = []
    h = same_pad(IMAGE_SIZE, 3, 2)[0]          # stem 3x3 s2 -> 112
    idx = 0
    for (rep, k, s, e, ci, co) in _STAGES:
        for r in range(rep):
            c_in = ci if r == 0 else co
            stride = s if r == 0 else 1
            h_out, lo, hi = same_pad(h, k, stride)
            out.append(Block(idx, k, stride, e, c_in, co, c_in * e,
                             max(1, int(c_in * SE_RATIO)), h, h_out, lo, hi,
                             stride == 1 and c_in == co))
            h = h_out
            idx += 1
    return out


BLOCKS = blocks()


def param_count() -> int:
    """Learnable parameters (conv/linear weights+biases and BN affine)."""
    n = 3 * 3 * 3 * STEM_OUT + 2 * STEM_OUT
    for b in BLOCKS:
        if b.expand != 1:
            n += b.c_in * b.c_exp + 2 * b.c_exp
        n += b.kernel * b.kernel * b.c_exp + 2 * b.c_exp
        n += b.c_exp * b.c_se + b.c_se + b.c_se * b.c_exp + b.c_exp
        n += b.c_exp * b.c_out + 2 * b.c_out
    n += BLOCKS[-1].c_out * HEAD_OUT + 2 * HEAD_OUT
    d = MLP_DIMS
    n += d[0] * d[1] + d[1] + 2 * d[1]
    n += d[1] * d[2] + d[2] + 2 * d[2]
    n += d[2] * d[3] + d[3]
    return n


def macs_per_image() -> dict:
    """Multiply-accumulates per 224x224 crop, split as SURVEY.md section 8 A7 does."""
    pw = dw = se = 0
    h_stem = BLOCKS[0].h_in
    stem = h_stem * h_stem * 27 * STEM_OUT
    for b in BLOCKS:
        if b.expand != 1:
            pw += b.h_in * b.h_in * b.c_in * b.c_exp
        dw += b.h_out * b.h_out * b.kernel * b.kernel * b.c_exp
        se += 2 * b.c_exp * b.c_se
        pw += b.h_out * b.h_out * b.c_exp * b.c_out
    hl = BLOCKS[-1].h_out
    head = hl * hl * BLOCKS[-1].c_out * HEAD_OUT
    d = MLP_DIMS
    mlp = d[0] * d[1] + d[1] * d[2] + d[2] * d[3]
    return {"stem": stem, "pointwise": pw + head, "depthwise": dw, "se": se, "mlp": mlp,
            "total": stem + pw + head + dw + se + mlp}


def depthwise_bytes_per_image(elem_bytes: int = 4) -> int:
    """Algorithmic HBM bytes of the 16 depthwise layers per crop: input + output
    activations + k*k*C weights (25.11 MB in fp32)."""
    n = 0
    for b in BLOCKS:
        n += (b.h_in * b.h_in + b.h_out * b.h_out + b.kernel * b.kernel) * b.c_exp
    return n * elem_bytes
