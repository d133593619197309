"""CPU checks of the MTCNN oracle (oracle/mtcnn_ref.py): what can be pinned is pinned (Pillow's resize),
the rest is checked for internal consistency (NMS against brute force, packing against the torch layout)."""
import numpy as np
import pytest
import torch

from oracle import mtcnn_ref as M
from tests import mt_images


@pytest.mark.parametrize("shape", [(300, 280), (97, 131), (160, 160), (641, 480), (50, 40), (161, 159), (1000, 37), (160, 90)])
def test_pil_resize_bit_exact_vs_pillow(shape):
    from PIL import Image

    a = np.random.RandomState(sum(shape)).randint(0, 256, shape + (3,)).astype(np.uint8)
    want = np.asarray(Image.fromarray(a).resize((160, 160), Image.BILINEAR))
    assert np.array_equal(M.pil_resize_bilinear(a, 160, 160), want)


def test_scale_pyramid_matches_closed_form():
    s = M.scale_pyramid(300, 280)
    assert len(s) == 8 and s[0] == 0.6
    assert all(abs(s[i + 1] / s[i] - 0.709) < 1e-12 for i in range(len(s) - 1))
    assert min(300, 280) * s[-1] >= 12 > min(300, 280) * s[-1] * 0.709
    assert M.scale_pyramid(19, 400) == []                    # 19 * 0.6 < 12: no level, no face


def _iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0]))
    ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def test_nms_iou_is_greedy_suppression():
    rs = np.random.RandomState(0)
    xy = rs.rand(200, 2) * 100
    wh = 10 + rs.rand(200, 2) * 30
    b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    sc = rs.rand(200).astype(np.float32)
    keep = M.nms_iou(b, sc, 0.5)
    assert list(sc[keep]) == sorted(sc[keep], reverse=True)
    for i, ki in enumerate(keep):
        for kj in keep[:i]:
            assert _iou(b[ki], b[kj]) <= 0.5 + 1e-6
    dropped = set(range(200)) - set(keep.tolist())
    for d in dropped:
        assert any(sc[k] >= sc[d] and _iou(b[k], b[d]) > 0.5 - 1e-6 for k in keep)


def test_dense_packing_matches_torch_flatten_order(pkg, mtcnn_sd):
    """pack_mtcnn_tensors re-orders dense4/dense5 inputs from the package's (W,H,C) flatten to NHWC."""
    t = pkg.weights.pack_mtcnn_tensors(mtcnn_sd)
    rs = np.random.RandomState(1)
    for net, name, c in (("rnet", "dense4", 64), ("onet", "dense5", 128)):
        x = rs.randn(2, c, 3, 3).astype(np.float32)                      # NCHW conv output
        w = torch.from_numpy(mtcnn_sd[f"{net}.{name}.weight"])
        want = torch.from_numpy(x).permute(0, 3, 2, 1).reshape(2, -1) @ w.T
        got = x.transpose(0, 2, 3, 1).reshape(2, -1) @ t[f"mtcnn.{net}.{name}.w"]
        assert np.allclose(got, want.numpy(), atol=1e-5)
    w = mtcnn_sd["pnet.conv2.weight"]
    assert np.array_equal(t["mtcnn.pnet.conv2.w"][3, 1, 2, :], w[:, 3, 1, 2])


def test_cascade_outcomes_and_determinism(pkg, mtcnn_sd):
    sd = pkg.weights.to_torch(mtcnn_sd)
    found = []
    for img in mt_images.images():
        taps = {}
        a = M.mtcnn_forward(sd, img, taps)
        b = M.mtcnn_forward(sd, img)
        assert (a is None) == (b is None)
        if a is not None:
            assert a.shape == (3, 160, 160) and a.dtype == np.float32 and np.array_equal(a, b)
            assert 0 <= a.min() and a.max() <= 255 and np.array_equal(a, np.round(a))
            sel = taps["selected"]
            assert sel[4] == taps["stage3"][:, 4].max() > 0.7
        found.append(a is not None)
    assert any(found) and not all(found)                     # both outcomes are exercised
    assert M.mtcnn_forward(sd, mt_images.textured(12, 40, 8)) is None
