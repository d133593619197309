"""CPU: cross-checks oracle/b0_ref.py's backbone against HuggingFace `transformers`' EfficientNet -
an independent implementation of the same published B0 architecture (same TF-SAME padding, BN eps
1e-3, SE from the un-expanded width) - by copying the seeded weights into it.  This does not pin
the oracle to the reference's dependency (efficientnet_pytorch is absent), but it rules out a
private misreading of the architecture."""
import numpy as np
import pytest
import torch

from oracle import b0_ref

transformers = pytest.importorskip("transformers")


def _hf_model():
    from transformers import EfficientNetConfig, EfficientNetModel

    cfg = EfficientNetConfig(width_coefficient=1.0, depth_coefficient=1.0, image_size=224, hidden_dim=1280,
                             dropout_rate=0.2, batch_norm_eps=1e-3)
    return EfficientNetModel(cfg).eval()


def _copy_weights(hf, sd):
    """HF parameter order follows the same layer order as efficientnet_pytorch; copy by position/shape."""
    t = {k[4:]: v for k, v in sd.items()}                       # strip 'net.'
    src = []

    def conv(name):
        src.append(t[name + ".weight"])

    def bn(name):
        src.extend([t[name + ".weight"], t[name + ".bias"]])

    conv("_conv_stem"); bn("_bn0")
    for i, (k, s, e, ci, co) in enumerate(b0_ref.block_list()):
        p = f"_blocks.{i}"
        if e != 1:
            conv(p + "._expand_conv"); bn(p + "._bn0")
        conv(p + "._depthwise_conv"); bn(p + "._bn1")
        src.extend([t[p + "._se_reduce.weight"], t[p + "._se_reduce.bias"], t[p + "._se_expand.weight"], t[p + "._se_expand.bias"]])
        conv(p + "._project_conv"); bn(p + "._bn2")
    conv("_conv_head"); bn("_bn1")
    params = [p for n, p in hf.named_parameters()]
    assert len(params) == len(src), (len(params), len(src))
    with torch.no_grad():
        for p, s in zip(params, src):
            assert tuple(p.shape) == tuple(s.shape), (tuple(p.shape), tuple(s.shape))
            p.copy_(s)
    # running statistics are buffers, in the same BN order
    means = [v for k, v in t.items() if k.endswith("running_mean") and not k.startswith("_fc")]
    vars_ = [v for k, v in t.items() if k.endswith("running_var") and not k.startswith("_fc")]
    hm = [b for n, b in hf.named_buffers() if n.endswith("running_mean")]
    hv = [b for n, b in hf.named_buffers() if n.endswith("running_var")]
    assert len(hm) == len(means)
    # our dict order is stem, per block (bn0, bn1, bn2), head - same as HF's module order
    for b, s in zip(hm, means):
        b.copy_(s)
    for b, s in zip(hv, vars_):
        b.copy_(s)


def test_backbone_equals_hf_efficientnet(pkg, seeded_sd):
    sd = pkg.weights.to_torch(seeded_sd)
    hf = _hf_model()
    try:
        _copy_weights(hf, sd)
    except AssertionError as e:
        pytest.skip(f"transformers' EfficientNet layout differs from the expected order: {e}")
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = hf(pixel_values=x).pooler_output
    got = b0_ref.extract_features(sd, x)
    assert got.shape == want.shape == (2, 1280)
    assert (got - want).abs().max() <= 1e-4, float((got - want).abs().max())
