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


@pytest.mark.parametrize("h,w,kw", [(256, 256, dict(quality=90)), (301, 217, dict(quality=85, subsampling=0)),
                                    (240, 320, dict(quality=60, subsampling=1)), (97, 131, dict(quality=95)),
                                    (1080, 1920, dict(quality=85))])
def test_speculative_chunks_give_the_sequential_result(pkg, monkeypatch, h, w, kw):
    """The parallel entropy decoder (speculative chunks stitched on matching (bit position, MCU slot) states, csrc/
    jpeg_decode.hip) must return the sequential decoder's coefficients whatever the number of chunks and wherever
    their borders fall - including chunk counts far above the pool size and chunks shorter than a block."""
    data = _jpeg(_img(h, w, 31), **kw)
    monkeypatch.setenv("DFD_JPEG_CHUNKS", "1")
    want = pkg._lib.jpeg_coefficients(data)["coef"]
    for chunks in (2, 3, 7, 16, 61, 256):
        monkeypatch.setenv("DFD_JPEG_CHUNKS", str(chunks))
        got = pkg._lib.jpeg_coefficients(data)["coef"]
        assert np.array_equal(got, want), f"{chunks} chunks"
    gray = _jpeg(_img(200, 333, 32)[..., 0], quality=80)
    monkeypatch.setenv("DFD_JPEG_CHUNKS", "1")
    want = pkg._lib.jpeg_coefficients(gray)["coef"]
    monkeypatch.setenv("DFD_JPEG_CHUNKS", "9")
    assert np.array_equal(pkg._lib.jpeg_coefficients(gray)["coef"], want)


def test_decoder_pool_survives_fork(pkg):
    """The entropy decoder's host threads do not exist in a fork()ed child (a pre-fork WSGI server): the child must
    decode on its own thread instead of waiting for workers that never come."""
    import os

    data = _jpeg(_img(480, 640, 41), quality=85)
    want = pkg._lib.jpeg_coefficients(data)["coef"]              # creates the pool in this process
    pid = os.fork()
    if pid == 0:
        try:
            ok = np.array_equal(pkg._lib.jpeg_coefficients(data)["coef"], want)
        except BaseException:
            ok = False
        os._exit(0 if ok else 3)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_truncated_scan_is_loud(pkg):
    """A file cut inside its entropy-coded data raises (the old byte-wise reader decoded zeros for the missing MCUs)."""
    data = _jpeg(_img(128, 128, 33), quality=90)
    cut = data[:len(data) * 2 // 3]
    with pytest.raises(pkg._lib.DfdError):
        pkg._lib.jpeg_coefficients(cut)


def _rewrite_first_dht(data: bytes, bits1: int) -> bytes:
    """the file with bits[1] of its first Huffman table set to `bits1` (the count of 1-bit codes: at most 2 fit)"""
    i = data.index(b"\xff\xc4")
    out = bytearray(data)
    out[i + 5] = bits1                                          # FF C4, length (2), Tc/Th (1), bits[1]
    return bytes(out)


def test_oversubscribed_huffman_tables_are_rejected(pkg):
    """code lengths that over-subscribe the code space (Kraft sum > 1) used to index past the 9-bit
    lookahead table on the caller's stack (bits[1] = 3 -> look[512..767]; bits[1] = 255 -> ~128 KB past it).  libjpeg
    (jdhuff.c) rejects them; so does this decoder - before any table entry is written."""
    good = _jpeg(_img(48, 64, 5), quality=85)
    pkg._lib.jpeg_coefficients(good)
    for bits1 in (3, 17, 255):
        with pytest.raises(pkg._lib.DfdError) as e:
            pkg._lib.jpeg_coefficients(_rewrite_first_dht(good, bits1))
        assert e.value.code == -1
    # a whole-table variant: every length claims 16 codes (total 256 passes the old `total <= 256` check)
    i = good.index(b"\xff\xc4")
    seg = bytes([0xFF, 0xC4, 0x01, 0x13, 0x00]) + bytes([16] * 16) + bytes(range(256))
    with pytest.raises(pkg._lib.DfdError):
        pkg._lib.jpeg_coefficients(good[:i] + seg + good[i:])


def test_header_only_giant_frames_are_refused_before_allocation(pkg):
    """a 600-byte file claiming 65535 x 65535 made decode_scan zero-fill tens of GB.  The pixel cap
    (2^26) answers DFD_ERR_UNSUPPORTED so the server's Pillow path (with its own MAX_IMAGE_PIXELS) decides."""
    good = bytearray(_jpeg(_img(48, 64, 6), quality=85))
    i = bytes(good).index(b"\xff\xc0")
    good[i + 5:i + 9] = bytes([0xFF, 0xFF, 0xFF, 0xFF])         # height, width = 65535
    with pytest.raises(pkg._lib.DfdError) as e:
        pkg._lib.jpeg_coefficients(bytes(good))
    assert e.value.code == -7


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
