"""Host mirror of the reference's ``model.py`` over the HIP classifier.

`DeepfakeEfficientNet` keeps the constructor and call surface of reference
model.py:21-102 (``forward(rgb, freq=None) -> (B,1)`` logits, ``extract_features ->
(B,1280)``, ``forward_with_projection``); the arithmetic runs in libdfd_hip.so
(`dfd_classify_nchw`, include/dfd_hip.h).  It is inference-only (the reference's
``.eval()`` path); there is no CPU implementation behind it.
"""
from __future__ import annotations

import logging
from typing import Dict, Mapping, Optional

import numpy as np

from . import b0_arch as A
from . import weights as W
from ._lib import DfdError, Handle

log = logging.getLogger(__name__)


class _LayerInfo:
    """Shape-only stand-in for one entry of the reference's ``net._fc`` Sequential."""

    def __init__(self, kind: str, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return f"{self.kind}({', '.join(f'{k}={v}' for k, v in self.__dict__.items() if k != 'kind')})"


def _fc_layout(dropout: float):
    d = A.MLP_DIMS
    return [
        _LayerInfo("Dropout", p=dropout),
        _LayerInfo("Linear", in_features=d[0], out_features=d[1]),
        _LayerInfo("BatchNorm1d", num_features=d[1]),
        _LayerInfo("ReLU"),
        _LayerInfo("Dropout", p=dropout * 0.7),
        _LayerInfo("Linear", in_features=d[1], out_features=d[2]),
        _LayerInfo("BatchNorm1d", num_features=d[2]),
        _LayerInfo("ReLU"),
        _LayerInfo("Dropout", p=dropout * 0.5),
        _LayerInfo("Linear", in_features=d[2], out_features=d[3]),
    ]


class _NetInfo:
    def __init__(self, dropout):
        self._fc = _fc_layout(dropout)      # reference model.py:50-61 (10 entries)
        self._conv_head = _LayerInfo("Conv2d", in_channels=320, out_channels=A.HEAD_OUT, kernel_size=1)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class DeepfakeEfficientNet:
    """EfficientNet-B0 + 1280->512->256->1 head, executed on one MI355X.

    Args mirror reference model.py:36: ``pretrained`` is accepted for signature
    parity (the reference fetches ImageNet weights there; this build has no
    network and starts from `weights.seeded_state_dict(seed)` until
    `load_state_dict` is called), ``dropout`` only shapes the reported head.
    """

    def __init__(self, pretrained: bool = True, dropout: float = 0.5, *, device: int = 0,
                 max_batch: int = 8, seed: int = 0, state_dict: Optional[Mapping[str, np.ndarray]] = None):
        self.net = _NetInfo(dropout)
        self.device_index = int(device)
        self.max_batch = int(max_batch)
        self.training = False
        self._handle: Optional[Handle] = None
        self._state: Dict[str, np.ndarray] = dict(state_dict) if state_dict is not None else W.seeded_state_dict(seed)
        if pretrained and state_dict is None:
            log.info("pretrained=True: no ImageNet weights offline; using seeded random init (seed=%d)", seed)

    # ---- torch.nn.Module look-alikes used by the reference call sites
    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def parameters_count(self) -> int:
        return A.param_count()

    def state_dict(self) -> Dict[str, np.ndarray]:
        return dict(self._state)

    def load_state_dict(self, state: Mapping, strict: bool = False):
        """reference deepfake_detection.py:44-59: returns (missing, unexpected) key lists."""
        new = {k: (v.detach().cpu().numpy() if _is_torch(v) else np.asarray(v)) for k, v in state.items()}
        want = set(self._state)
        missing = sorted(k for k in want - set(new) if not k.endswith("num_batches_tracked"))
        unexpected = sorted(set(new) - want)
        if strict and (missing or unexpected):
            raise KeyError(f"missing={missing[:5]} unexpected={unexpected[:5]}")
        for k in want & set(new):
            if new[k].shape != self._state[k].shape:
                raise ValueError(f"shape mismatch for {k}: {new[k].shape} vs {self._state[k].shape}")
            self._state[k] = new[k]
        if self._handle is not None:
            self._handle.close()
            self._handle = None
        return missing, unexpected

    # ---- device
    @property
    def handle(self) -> Handle:
        if self._handle is None:
            self._handle = Handle(W.pack_b0(self._state), device=self.device_index, max_batch=self.max_batch)
        return self._handle

    def _run(self, x, fn):
        torch_in = _is_torch(x)
        a = x.detach().cpu().numpy() if torch_in else np.asarray(x)
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.ndim != 4 or a.shape[1:] != (3, A.IMAGE_SIZE, A.IMAGE_SIZE):
            raise ValueError(f"expected (B,3,224,224), got {a.shape}")
        outs = []
        for i in range(0, a.shape[0], self.max_batch):
            outs.append(fn(a[i:i + self.max_batch]))
        out = np.concatenate(outs, axis=0)
        if torch_in:
            import torch

            return torch.from_numpy(out)
        return out

    def forward(self, rgb_input, freq_input=None):
        """(B,3,224,224) normalised RGB -> (B,1) logits; ``freq_input`` ignored (reference model.py:63-72)."""
        return self._run(rgb_input, self.handle.classify)

    __call__ = forward

    def extract_features(self, rgb_input):
        """(B,1280) pooled backbone features (reference model.py:74-89)."""
        return self._run(rgb_input, self.handle.extract_features)

    def forward_with_projection(self, rgb_input, freq_input=None):
        """reference model.py:91-98"""
        return self.forward(rgb_input), None

    def get_feature_extractor(self):
        """reference model.py:100-102 (GradCAM hook target; descriptor only here)."""
        return self.net._conv_head


def compute_frequency_features(image_bgr_or_rgb, size: int = 224, *, handle: Optional[Handle] = None):
    """FFT log-magnitude + DCT features, (2, size, size) float32 in [0,1] (reference model.py:105-149),
    computed on the GPU.  Only ``size=224`` (the one value the reference ever passes,
    deepfake_detection.py:392) is built."""
    if size != 224:
        raise ValueError("compute_frequency_features is built for size=224 only")
    if handle is None:
        from . import runtime

        handle = runtime.default_handle()
    return handle.frequency_features(image_bgr_or_rgb)


__all__ = ["DeepfakeEfficientNet", "DfdError", "compute_frequency_features"]
