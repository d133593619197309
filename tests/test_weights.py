"""CPU: architecture table, seeded generator, BN folding and blob packing."""
import struct

import numpy as np
import torch

from oracle import b0_ref


def test_param_and_mac_counts(pkg):
    A = pkg.b0_arch
    assert A.param_count() == 4_796_541
    macs = A.macs_per_image()
    assert macs["total"] == 385_321_440 and macs["depthwise"] == 34_532_064
    assert abs(A.depthwise_bytes_per_image() / 1e6 - 25.11) < 0.01
    assert len(A.BLOCKS) == 16
    assert [(b.pad_lo, b.pad_hi) for b in A.BLOCKS if b.stride == 2] == [(0, 1), (1, 2), (0, 1), (1, 2)]


def test_state_dict_matches_reference_naming(pkg, seeded_sd):
    learnable = sum(v.size for k, v in seeded_sd.items()
                    if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert learnable == pkg.b0_arch.param_count()
    for k in ("net._conv_stem.weight", "net._bn0.running_var", "net._blocks.0._depthwise_conv.weight",
              "net._blocks.15._se_expand.bias", "net._conv_head.weight", "net._fc.1.weight",
              "net._fc.2.running_mean", "net._fc.9.bias"):
        assert k in seeded_sd
    assert seeded_sd["net._blocks.1._expand_conv.weight"].shape == (96, 16, 1, 1)
    assert seeded_sd["net._blocks.3._depthwise_conv.weight"].shape == (144, 1, 5, 5)
    assert seeded_sd["net._blocks.1._se_reduce.weight"].shape == (4, 96, 1, 1)
    assert "net._blocks.0._expand_conv.weight" not in seeded_sd      # expand ratio 1
    again = pkg.weights.seeded_state_dict(0)
    assert all(np.array_equal(seeded_sd[k], again[k]) for k in seeded_sd)


def test_bn_folding_is_exact_enough(pkg, seeded_sd):
    """Folded stem (conv*a + b) equals conv -> BN(eps 1e-3) on the CPU to fp32 rounding."""
    t = pkg.weights.pack_b0_tensors(seeded_sd)
    tsd = pkg.weights.to_torch(seeded_sd)
    x = torch.randn(1, 3, 32, 32)
    want = torch.nn.functional.batch_norm(
        torch.nn.functional.conv2d(x, tsd["net._conv_stem.weight"]),
        tsd["net._bn0.running_mean"], tsd["net._bn0.running_var"], tsd["net._bn0.weight"], tsd["net._bn0.bias"],
        False, 0.0, 1e-3)
    w = torch.from_numpy(t["stem.w"]).permute(3, 2, 0, 1).contiguous()
    got = torch.nn.functional.conv2d(x, w, torch.from_numpy(t["stem.b"]))
    assert (got - want).abs().max() < 1e-5


def test_blob_roundtrip(pkg, seeded_sd):
    t = pkg.weights.pack_b0_tensors(seeded_sd)
    blob = pkg.weights.serialize(t)
    assert blob[:4] == b"DFDW"
    ver, cnt = struct.unpack("<II", blob[4:12])
    assert ver == 1 and cnt == len(t)
    entry = struct.Struct("<48sI4IQQ")
    for i, name in enumerate(t):
        nm, nd, d0, d1, d2, d3, off, nb = entry.unpack_from(blob, 12 + i * entry.size)
        assert nm.rstrip(b"\0").decode() == name and off % 64 == 0
        a = np.frombuffer(blob, np.float32, nb // 4, off)
        assert np.array_equal(a, t[name].ravel())


def test_load_state_dict_surface(pkg, seeded_sd):
    m = pkg.model.DeepfakeEfficientNet(pretrained=False)
    missing, unexpected = m.load_state_dict({**seeded_sd, "extra.key": np.zeros(1)}, strict=False)
    assert missing == [] and unexpected == ["extra.key"]
    missing, _ = m.load_state_dict({k: v for k, v in seeded_sd.items() if k != "net._fc.9.bias"})
    assert missing == ["net._fc.9.bias"]
