"""JPEG decode at the HTTP edge (reference backend_server.py:139-145 cv2.imdecode).
CPU: the library's host half (markers + Huffman decoding, csrc/jpeg_decode.hip) is pinned against libjpeg itself -
its coefficients pushed through the oracle's IDCT / upsampling / colour conversion must reproduce Pillow's decode bit
for bit, over sampling modes, odd sizes, qualities, optimised tables and restart intervals.
GPU: dfd_decode_jpeg (device half) == Pillow bit for bit; dfd_analyze_jpeg == dfd_analyze_frame on the decoded frame."""
import io

import numpy as np
import pytest
from PIL import Image

import frames as F
from oracle import jpeg_ref


def _img(h, w, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 90 * np.sin(xx / (7.0 + c) + c) * np.cos(yy / (11.0 - c)) for c in range(3)], -1)
    return np.clip(base + rs.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)


def _jpeg(bgr, **kw):
    buf = io.BytesIO()
    im = Image.fromarray(np.ascontiguousarray(bgr[..., ::-1])) if bgr.ndim == 3 else Image.fromarray(bgr)
    im.save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _pil_bgr(data):
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1])


CASES = [  # (h, w, save kwargs)
    (64, 64, dict(quality=90)),
    (50, 70, dict(quality=85)),                               # 4:2:0, neither dimension a multiple of 16
    (33, 17, dict(quality=75, subsampling=0)),                # 4:4:4, odd sizes
    (41, 95, dict(quality=60, subsampling=1)),                # 4:2:2 (h2v1)
    (120, 160, dict(quality=95, optimize=True)),              # optimised Huffman tables
    (97, 131, dict(quality=30)),                              # coarse quantisation: long zero runs, EOBs
    (1, 1, dict(quality=90)),
    (8, 9, dict(quality=90)),
    (270, 480, dict(quality=85)),                             # a quarter-scale 1080p frame: 270 is not a multiple of 16
]


@pytest.mark.parametrize("h,w,kw", CASES)
def test_host_entropy_decoder_pinned_to_libjpeg(pkg, h, w, kw):
    data = _jpeg(_img(h, w, h * 1000 + w), **kw)
    info = pkg._lib.jpeg_coefficients(data)
    assert (info["width"], info["height"]) == (w, h)
    got = jpeg_ref.decode_from_coefficients(info)
    assert np.array_equal(got, _pil_bgr(data))


def test_gray_and_restart_intervals(pkg):
    g = _img(45, 77, 3)[..., 1]
    data = _jpeg(g, quality=88)
    info = pkg._lib.jpeg_coefficients(data)
    assert info["components"] == 1
    assert np.array_equal(jpeg_ref.decode_from_coefficients(info), _pil_bgr(data))
    # restart markers: Pillow writes DRI when asked for restart_marker_blocks (libjpeg restart_interval)
    data = _jpeg(_img(64, 96, 5), quality=80, restart_marker_blocks=3)
    assert b"\xff\xdd" in data
    info = pkg._lib.jpeg_coefficients(data)
    assert np.array_equal(jpeg_ref.decode_from_coefficients(info), _pil_bgr(data))


def test_unsupported_and_garbage_are_loud(pkg):
    prog = _jpeg(_img(40, 40, 1), quality=80, progressive=True)
    with pytest.raises(pkg._lib.DfdError) as e:
        pkg._lib.jpeg_coefficients(prog)
    assert e.value.code == -7                                  # DFD_ERR_UNSUPPORTED: the host keeps its own decoder
    for junk in (b"", b"not a jpeg", _jpeg(_img(32, 32, 2))[:200]):
        with pytest.raises(pkg._lib.DfdError):
            pkg._lib.jpeg_coefficients(junk)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,kw", CASES + [(1080, 1920, dict(quality=85))])
def test_device_decode_equals_libjpeg(b0_handle, h, w, kw):
    data = _jpeg(_img(h, w, h * 1000 + w), **kw)
    assert np.array_equal(b0_handle.decode_jpeg(data), _pil_bgr(data))


@pytest.mark.gpu
def test_analyze_jpeg_equals_analyze_frame(pkg, b0_handle):
    h = b0_handle
    for i, frame in enumerate((F.natural_like(480, 640, seed=9), F.face_frame(640, 480, 2), F.blank_frame(640, 480))):
        data = _jpeg(frame, quality=85)
        decoded = _pil_bgr(data)
        h.forensics_reset(810 + i)
        h.forensics_reset(820 + i)
        a = h.analyze_frame(decoded, True, stream_id=810 + i, max_faces=4)
        b = h.analyze_jpeg(data, True, stream_id=820 + i, max_faces=4)
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3]) and b[4] == (480, 640)
    with pytest.raises(pkg._lib.DfdError) as e:
        h.decode_jpeg(_jpeg(_img(40, 40, 1), progressive=True))
    assert e.value.code == h.UNSUPPORTED
