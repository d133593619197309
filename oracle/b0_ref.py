"""ORACLE (test infrastructure, never shipped or measured as the product):
CPU restatement of the reference classifier forward, torch-CPU fp32.

Follows reference model.py:21-72 (``DeepfakeEfficientNet.forward`` == ``self.net(x)``
with ``net._fc`` replaced by Dropout/Linear/BN1d/ReLU x2 + Linear, model.py:50-61)
and, for the backbone arithmetic, the published ``efficientnet_pytorch`` B0
(lukemelas; imported at reference model.py:18 but NOT vendored and NOT pinned in
reference requirements.txt): TF-"SAME" static padding for image_size 224, BN
eps 1e-3, swish = x*sigmoid(x), SE squeeze width max(1,int(c_in*0.25)) with biased
1x1 convs, residual only when stride 1 and c_in == c_out, eval-mode dropout and
drop-connect are identities.

PARITY UNPINNED for logits: the reference tests pin only shapes/ranges/determinism
for this path (reference tests/test_functional.py:93-110, test_reliability.py:123-132)
and neither the reference module nor its dependency can be imported here
(ModuleNotFoundError: cv2 / efficientnet_pytorch).  Pinned structure: 10-entry head
1280->512->256->1 (tests/test_functional.py:70-79), (B,1) output, <8M parameters
(tests/test_performance.py:234-241).  `tests/test_oracle_b0.py` additionally checks
this restatement against the independent HuggingFace `transformers` EfficientNet
implementation of the same published architecture.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional

import torch
import torch.nn.functional as F

# (repeats, kernel, stride, expand, in, out) - efficientnet_pytorch B0 block args
_B0 = ((1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
       (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320))
_BN_EPS = 1e-3


def _same_conv(x, w, stride, groups=1, bias=None):
    """Conv2dStaticSamePadding: zero-pad (lo = total//2, hi = total - lo) then VALID conv."""
    k = w.shape[-1]
    ih, iw = x.shape[-2:]
    oh, ow = -(-ih // stride), -(-iw // stride)
    ph = max((oh - 1) * stride + k - ih, 0)
    pw = max((ow - 1) * stride + k - iw, 0)
    if ph or pw:
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return F.conv2d(x, w, bias, stride=stride, groups=groups)


def _bn(x, sd, p, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _swish(x):
    return x * torch.sigmoid(x)


def block_list():
    out = []
    for (rep, k, s, e, ci, co) in _B0:
        for r in range(rep):
            out.append((k, s if r == 0 else 1, e, ci if r == 0 else co, co))
    return out


def extract_features(sd: Mapping[str, torch.Tensor], x: torch.Tensor,
                     taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """Backbone up to the global average pool -> (B,1280); reference model.py:74-89."""
    sd = {(k if k.startswith("net.") else "net." + k): v for k, v in sd.items()}
    x = _swish(_bn(_same_conv(x, sd["net._conv_stem.weight"], 2), sd, "net._bn0", _BN_EPS))
    if taps is not None:
        taps["stem"] = x
    for i, (k, s, e, ci, co) in enumerate(block_list()):
        p = f"net._blocks.{i}"
        inp = x
        if e != 1:
            x = _swish(_bn(_same_conv(x, sd[p + "._expand_conv.weight"], 1), sd, p + "._bn0", _BN_EPS))
            if taps is not None:
                taps[f"b{i}.exp"] = x
        x = _swish(_bn(_same_conv(x, sd[p + "._depthwise_conv.weight"], s, groups=x.shape[1]),
                       sd, p + "._bn1", _BN_EPS))
        if taps is not None:
            taps[f"b{i}.dw"] = x
        q = F.adaptive_avg_pool2d(x, 1)
        q = _swish(F.conv2d(q, sd[p + "._se_reduce.weight"], sd[p + "._se_reduce.bias"]))
        q = F.conv2d(q, sd[p + "._se_expand.weight"], sd[p + "._se_expand.bias"])
        if taps is not None:
            taps[f"b{i}.gate"] = torch.sigmoid(q)
        x = torch.sigmoid(q) * x
        x = _bn(_same_conv(x, sd[p + "._project_conv.weight"], 1), sd, p + "._bn2", _BN_EPS)
        if s == 1 and ci == co:
            x = x + inp
        if taps is not None:
            taps[f"b{i}.out"] = x
    x = _swish(_bn(_same_conv(x, sd["net._conv_head.weight"], 1), sd, "net._bn1", _BN_EPS))
    if taps is not None:
        taps["head"] = x
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def head(sd: Mapping[str, torch.Tensor], f: torch.Tensor) -> torch.Tensor:
    """Eval-mode ``net._fc`` Sequential (reference model.py:50-61); dropouts are identities."""
    sd = {(k if k.startswith("net.") else "net." + k): v for k, v in sd.items()}
    f = F.linear(f, sd["net._fc.1.weight"], sd["net._fc.1.bias"])
    f = F.relu(F.batch_norm(f, sd["net._fc.2.running_mean"], sd["net._fc.2.running_var"],
                            sd["net._fc.2.weight"], sd["net._fc.2.bias"], False, 0.0, 1e-5))
    f = F.linear(f, sd["net._fc.5.weight"], sd["net._fc.5.bias"])
    f = F.relu(F.batch_norm(f, sd["net._fc.6.running_mean"], sd["net._fc.6.running_var"],
                            sd["net._fc.6.weight"], sd["net._fc.6.bias"], False, 0.0, 1e-5))
    return F.linear(f, sd["net._fc.9.weight"], sd["net._fc.9.bias"])


@torch.no_grad()
def forward(sd: Mapping[str, torch.Tensor], x: torch.Tensor,
            taps: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """``DeepfakeEfficientNet.forward`` (reference model.py:63-72): (B,3,224,224) -> (B,1) logits."""
    return head(sd, extract_features(sd, x.float(), taps))


def sigmoid_prob(logit: torch.Tensor) -> torch.Tensor:
    """reference deepfake_detection.py:397-398"""
    return torch.sigmoid(logit)
