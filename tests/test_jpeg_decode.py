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


# ---- entropy decoding on the device (round 4, csrc/jpeg_gpu_entropy.h): the batch path -------------------------------
def _noise(h, w, seed):
    return np.random.default_rng(seed).integers(50, 200, (h, w, 3), dtype=np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,kw", CASES + [(1080, 1920, dict(quality=85)), (200, 333, dict(quality=92, optimize=True, subsampling=0))])
def test_device_entropy_batch_equals_libjpeg(b0_handle, h, w, kw):
    """three different files of one size per batch: every frame == Pillow's decode, and the device decoder did them"""
    datas = [_jpeg(_img(h, w, h * 1000 + w + i), **kw) for i in range(3)]
    b0_handle.set_option("jpeg_device_entropy", 1)                  # always (the default sends batches under 1 MiB to the host pool)
    try:
        d0, h0 = b0_handle.jpeg_decode_counts()
        got = b0_handle.decode_jpeg_batch(datas)
        d1, h1 = b0_handle.jpeg_decode_counts()
    finally:
        b0_handle.set_option("jpeg_device_entropy", 2)
    for i, data in enumerate(datas):
        assert np.array_equal(got[i], _pil_bgr(data)), i
    # (the size query decodes nothing; the second call decodes the three frames)
    assert (d1 - d0, h1 - h0) == (3, 0)


@pytest.mark.gpu
def test_device_entropy_worst_case_texture_and_gray(b0_handle):
    """random texture at 1080p (1.2 MB of entropy-coded data per frame, 2,400 lanes each), gray files, and a batch whose
    frames carry their own optimised tables"""
    datas = [_jpeg(_noise(1080, 1920, 70 + i), quality=85) for i in range(4)]
    d0, h0 = b0_handle.jpeg_decode_counts()
    got = b0_handle.decode_jpeg_batch(datas)                        # 5 MB of scans: the default picks the device
    assert b0_handle.jpeg_decode_counts() == (d0 + 4, h0)
    for i, data in enumerate(datas):
        assert np.array_equal(got[i], _pil_bgr(data)), i
    b0_handle.set_option("jpeg_device_entropy", 1)
    gray = [_jpeg(_img(200, 333, 32 + i)[..., 0], quality=80) for i in range(2)]
    got = b0_handle.decode_jpeg_batch(gray)
    for i, data in enumerate(gray):
        assert np.array_equal(got[i], _pil_bgr(data)), i
    opt = [_jpeg(_noise(240, 320, 5 + i) // (i + 1), quality=70 + 5 * i, optimize=True) for i in range(5)]
    got = b0_handle.decode_jpeg_batch(opt)
    for i, data in enumerate(opt):
        assert np.array_equal(got[i], _pil_bgr(data)), i
    b0_handle.set_option("jpeg_device_entropy", 2)
    small = [_jpeg(_img(120, 160, 50 + i), quality=85) for i in range(3)]
    d0, h0 = b0_handle.jpeg_decode_counts()
    got = b0_handle.decode_jpeg_batch(small)                        # 30 KB: the default leaves it to the host pool
    assert b0_handle.jpeg_decode_counts() == (d0, h0 + 3) and all(np.array_equal(got[i], _pil_bgr(small[i])) for i in range(3))


@pytest.mark.gpu
def test_device_entropy_result_does_not_depend_on_the_chunk_size(b0_handle):
    datas = [_jpeg(_img(480, 640, 90 + i), quality=85) for i in range(2)] + [_jpeg(_noise(480, 640, 3), quality=95)]
    want = [_pil_bgr(d) for d in datas]
    b0_handle.set_option("jpeg_device_entropy", 1)
    try:
        for chunk in (256, 512, 1024, 4096, 65536):
            b0_handle.set_option("jpeg_chunk_bytes", chunk)
            got = b0_handle.decode_jpeg_batch(datas)
            assert all(np.array_equal(g, w) for g, w in zip(got, want)), chunk
    finally:
        b0_handle.set_option("jpeg_chunk_bytes", 512)
        b0_handle.set_option("jpeg_device_entropy", 2)


@pytest.mark.gpu
def test_device_entropy_leaves_what_it_cannot_vouch_for_to_the_host_decoder(pkg, b0_handle):
    """restart-interval files and batches of mixed sampling take the host path (same bits); a truncated scan is an error
    from whichever decoder meets it; the option switches the device decoder off"""
    h = b0_handle
    h.set_option("jpeg_device_entropy", 1)
    rst = [_jpeg(_img(64, 96, 5 + i), quality=80, restart_marker_blocks=3) for i in range(2)]
    d0, h0 = h.jpeg_decode_counts()
    got = h.decode_jpeg_batch(rst)
    d1, h1 = h.jpeg_decode_counts()
    assert all(np.array_equal(got[i], _pil_bgr(rst[i])) for i in range(2)) and (d1 - d0, h1 - h0) == (0, 2)
    mixed = [_jpeg(_img(64, 96, 1), quality=80), _jpeg(_img(64, 96, 2), quality=80, subsampling=0)]
    got = h.decode_jpeg_batch(mixed)
    assert all(np.array_equal(got[i], _pil_bgr(mixed[i])) for i in range(2))
    good = _jpeg(_img(128, 128, 33), quality=90)
    cut = good[: len(good) // 2] + b"\xff\xd9"
    with pytest.raises(pkg._lib.DfdError):
        h.decode_jpeg_batch([good, cut])
    try:
        h.set_option("jpeg_device_entropy", 0)
        d0, h0 = h.jpeg_decode_counts()
        datas = [_jpeg(_img(120, 160, 7 + i), quality=85) for i in range(2)]
        got = h.decode_jpeg_batch(datas)
        d1, h1 = h.jpeg_decode_counts()
        assert all(np.array_equal(got[i], _pil_bgr(datas[i])) for i in range(2)) and (d1 - d0, h1 - h0) == (0, 2)
    finally:
        h.set_option("jpeg_device_entropy", 2)


@pytest.mark.gpu
def test_analyze_jpegs_host_equals_decoding_first(pkg, seeded_sd):
    """dfd_analyze_jpegs_host (scans uploaded chunk by chunk, decoded on the device, analysed) == Pillow decode of the same
    files + dfd_analyze_batch_device: boxes, logits, forensic probabilities; pinned and pageable input; a ragged last chunk"""
    W = pkg.weights
    h = pkg._lib.Handle(W.pack_all(seeded_sd, W.seeded_ssd_state_dict(0)), device=0, max_batch=16)
    try:
        frames = [F.natural_like(270, 480, seed=60 + i) for i in range(5)] + [F.face_frame(480, 270, 3)] + [_noise(270, 480, 9)]
        datas = [_jpeg(f, quality=85) for f in frames]
        decoded = np.stack([_pil_bgr(d) for d in datas])
        boxes = [[(40, 30, 120, 140), (250, 60, 160, 180)]] * len(datas)
        fd = h.alloc(decoded.nbytes).upload(decoded)
        want = h.analyze_batch_device(fd.ptr, len(datas), 270, 480, forced_boxes=boxes, max_faces=2, with_forensics=True)
        fd.free()
        packed = h.pack_jpegs(datas)
        for kw in (dict(packed=packed), dict()):
            got = h.analyze_jpegs_host(datas, 3, forced_boxes=boxes, max_faces=2, with_forensics=True, **kw)
            assert got[3] == (270, 480) and got[0] == want[0]
            assert all(np.array_equal(a, b) for a, b in zip(got[1], want[1]))
            assert np.array_equal(got[2], want[2])
        h.host_free(packed[0])
        with pytest.raises(pkg._lib.DfdError) as e:
            h.analyze_jpegs_host([_jpeg(frames[0], quality=85, restart_marker_blocks=2)] * 2, 2, forced_boxes=boxes[:2], max_faces=2)
        assert e.value.code == h.UNSUPPORTED
    finally:
        h.close()


@pytest.mark.gpu
def test_analyze_jpegs_host_fails_loudly_and_leaves_the_handle_usable(pkg, seeded_sd):
    """a truncated scan in the THIRD chunk (the producer thread is two chunks ahead by then): the call reports the host
    decoder's error, nothing hangs, and the next call on the same handle gives the right answer; a file of another size
    is refused before anything is queued"""
    W = pkg.weights
    h = pkg._lib.Handle(W.pack_all(seeded_sd, W.seeded_ssd_state_dict(0)), device=0, max_batch=16)
    try:
        good = [_jpeg(F.natural_like(270, 480, seed=80 + i), quality=85) for i in range(4)]
        boxes = [[(40, 30, 120, 140)]] * 8
        datas = [good[i % 4] for i in range(8)]
        want = h.analyze_jpegs_host(datas, 2, forced_boxes=boxes, max_faces=1)
        bad = list(datas)
        bad[5] = good[1][: len(good[1]) // 2] + b"\xff\xd9"
        with pytest.raises(pkg._lib.DfdError):
            h.analyze_jpegs_host(bad, 2, forced_boxes=boxes, max_faces=1)
        other = list(datas)
        other[6] = _jpeg(F.natural_like(240, 320, seed=3), quality=85)
        with pytest.raises(pkg._lib.DfdError):
            h.analyze_jpegs_host(other, 2, forced_boxes=boxes, max_faces=1)
        again = h.analyze_jpegs_host(datas, 2, forced_boxes=boxes, max_faces=1)
        assert again[0] == want[0] and all(np.array_equal(a, b) for a, b in zip(again[1], want[1]))
    finally:
        h.close()
