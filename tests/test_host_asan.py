"""Host-only code of libdfd_hip.so under AddressSanitizer + UBSan.

`make -C csrc asan-host` compiles the halves that read bytes nobody here wrote - JPEG markers, Huffman tables, the
entropy decoder with its speculative chunks and stitcher (csrc/jpeg_entropy.h = the body of dfd_jpeg_coefficients and of
every JPEG request, reference backend_server.py:139-145), the weights-blob table (csrc/blob_reader.h), the detectors'
integer box logic (csrc/host_boxes.h, reference face_detection.py:84-123) - as plain C++ with
-fsanitize=address,undefined -fno-sanitize-recover=all into csrc/../host_asan_driver.  A memory error or undefined
operation aborts the driver (non-zero exit + a sanitizer report); every case below must exit 0.  CPU only - the GPU
box never runs this."""
import io
import os
import struct
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-video-deepfake-detection_amd", "csrc")
DRIVER = os.path.join(ROOT, "real-time-video-deepfake-detection_amd", "host_asan_driver")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def driver():
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ with sanitizer runtimes in this image")
    subprocess.run(["make", "-C", CSRC, "asan-host"], check=True, capture_output=True)

    def run(*args, expect_ok=True):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        env.pop("DFD_JPEG_CHUNKS", None)
        r = subprocess.run([DRIVER, *map(str, args)], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, f"sanitizer report or crash for {args}:\n{r.stderr[-3000:]}"
        return r.stdout.strip()
    return run


def _img(h, w, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 90 * np.sin(xx / (7.0 + c) + c) * np.cos(yy / (11.0 - c)) for c in range(3)], -1)
    return np.clip(base + rs.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)


def _jpeg(rgb, **kw):
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _fields(line):
    return dict(kv.split("=") for kv in line.split())


def crafted_one_bit_tables(scan=b"\x00"):
    """both Huffman tables hold a single 1-bit code - DC symbol 0, AC symbol 0x0F (run 0, 15 value
    bits) - and the scan is one byte.  Every AC coefficient then consumes 16 bits of whatever follows the payload: the old
    decoder checked the bit position only between blocks and read its 8-byte windows ~120 bytes past a 17-byte buffer."""
    def seg(marker, body):
        return bytes([0xFF, marker]) + struct.pack(">H", len(body) + 2) + body
    dqt = seg(0xDB, bytes([0]) + bytes([1] * 64))
    sof = seg(0xC0, bytes([8]) + struct.pack(">HH", 8, 8) + bytes([1, 1, 0x11, 0]))
    dht_dc = seg(0xC4, bytes([0x00]) + bytes([1] + [0] * 15) + bytes([0]))
    dht_ac = seg(0xC4, bytes([0x10]) + bytes([1] + [0] * 15) + bytes([0x0F]))
    sos = seg(0xDA, bytes([1, 1, 0x00, 0, 63, 0]))
    return b"\xff\xd8" + dqt + sof + dht_dc + dht_ac + sos + scan + b"\xff\xd9"


def test_crafted_one_bit_tables_stay_inside_the_scan(driver, tmp_path, pkg):
    for k, scan in enumerate((b"\x00", b"", b"\x00" * 7, b"\x00" * 126, b"\x00" * 127)):
        p = tmp_path / f"onebit{k}.jpg"
        p.write_bytes(crafted_one_bit_tables(scan))
        out = _fields(driver("jpeg", p))
        # 1 + 63 * 16 bits = 127 bytes make the block complete; anything shorter is a truncated scan, loudly
        assert int(out["rc"]) == (0 if len(scan) >= 127 else -1), (len(scan), out)
    # the same bytes through the product library
    with pytest.raises(pkg._lib.DfdError):
        pkg._lib.jpeg_coefficients(crafted_one_bit_tables(b"\x00"))
    assert pkg._lib.jpeg_coefficients(crafted_one_bit_tables(b"\x00" * 127))["coef"].size == 64


CASES = [(64, 64, dict(quality=90)), (50, 70, dict(quality=85)), (33, 17, dict(quality=75, subsampling=0)),
         (41, 95, dict(quality=60, subsampling=1)), (120, 160, dict(quality=95, optimize=True)), (1, 1, dict(quality=90)),
         (270, 480, dict(quality=85))]


def test_good_files_decode_clean_and_independent_of_chunking(driver, tmp_path, pkg):
    for i, (h, w, kw) in enumerate(CASES):
        data = _jpeg(_img(h, w, 7 * i + 1)[..., ::-1].copy(), **kw)
        p = tmp_path / f"good{i}.jpg"
        p.write_bytes(data)
        base = _fields(driver("jpeg", p))
        assert base["rc"] == "0" and int(base["count"]) == pkg._lib.jpeg_coefficients(data)["coef"].size
        for chunks in (1, 2, 7, 64, 256):
            assert _fields(driver("jpeg", p, chunks)) == base, (i, chunks)
    # restart intervals: independent segments
    im = Image.fromarray(_img(96, 128, 5))
    buf = io.BytesIO()
    try:
        im.save(buf, format="JPEG", quality=85, restart_marker_blocks=3)
    except TypeError:
        pytest.skip("this Pillow cannot write restart markers")
    p = tmp_path / "rst.jpg"
    p.write_bytes(buf.getvalue())
    assert _fields(driver("jpeg", p))["rc"] == "0"


def test_truncations_and_header_edits_are_rejected_without_memory_errors(driver, tmp_path):
    data = _jpeg(_img(128, 128, 33), quality=90)
    sos = data.index(b"\xff\xda")
    for k, cut in enumerate([3, 20, sos - 1, sos + 5, sos + 14, sos + 15, sos + 40, len(data) // 2, len(data) * 2 // 3, len(data) - 3]):
        p = tmp_path / f"cut{k}.jpg"
        p.write_bytes(data[:cut])
        assert int(_fields(driver("jpeg", p))["rc"]) < 0, cut
    # over-subscribed code lengths and a 65535 x 65535 header
    i = data.index(b"\xff\xc4")
    for k, bits1 in enumerate((3, 17, 255)):
        bad = bytearray(data)
        bad[i + 5] = bits1
        p = tmp_path / f"dht{k}.jpg"
        p.write_bytes(bytes(bad))
        assert _fields(driver("jpeg", p))["rc"] == "-1"
    big = bytearray(data)
    j = data.index(b"\xff\xc0")
    big[j + 5:j + 9] = b"\xff\xff\xff\xff"
    p = tmp_path / "big.jpg"
    p.write_bytes(bytes(big))
    assert _fields(driver("jpeg", p))["rc"] == "-7"


def test_seeded_mutations_of_jpegs(driver, tmp_path):
    """4,000 seeded mutations (byte flips biased to the headers, FF insertions, truncations; 1 / 7 / 64 speculative
    chunks) of three files in-process: the decoder may accept or reject, it may not touch memory it does not own."""
    total = 0
    for k, (h, w, kw, n) in enumerate([(48, 64, dict(quality=85), 2000), (40, 56, dict(quality=50, subsampling=0), 1000),
                                       (270, 480, dict(quality=85), 1000)]):
        p = tmp_path / f"fz{k}.jpg"
        p.write_bytes(_jpeg(_img(h, w, 11 + k), **kw))
        out = _fields(driver("jpegfuzz", p, 1234 + k, n))
        assert int(out["ok"]) + int(out["rejected"]) == n and int(out["rejected"]) > n // 10
        total += n
    assert total == 4000


def test_blob_reader_and_its_mutations(driver, tmp_path, pkg):
    W = pkg.weights
    blob = W.serialize({"a.w": np.arange(24, dtype=np.float32).reshape(2, 3, 4), "b": np.ones(5, np.float32)})
    p = tmp_path / "w.blob"
    p.write_bytes(bytes(blob))
    out = _fields(driver("blob", p))
    assert out["rc"] == "0" and out["tensors"] == "2" and float(out["sum"]) == 24 * 23 / 2 + 5
    for k, cut in enumerate((0, 3, 11, 12, 40, 96, len(blob) // 2)):
        q = tmp_path / f"wc{k}.blob"
        q.write_bytes(bytes(blob)[:cut])
        assert driver("blob", q).startswith("rc=-2")
    # extents whose 64-bit product wraps onto the stored byte count
    bad = bytearray(blob)
    e = 12
    bad[e + 48:e + 52] = struct.pack("<I", 4)
    bad[e + 52:e + 68] = struct.pack("<IIII", 0x80000000, 0x80000000, 4, 6)       # product = 24 * 2^64 -> wraps to 0... x 24
    q = tmp_path / "wrap.blob"
    q.write_bytes(bytes(bad))
    assert driver("blob", q).startswith("rc=-2")
    out = _fields(driver("blobfuzz", p, 99, 3000))
    assert int(out["ok"]) + int(out["rejected"]) == 3000


def test_detector_box_logic_on_hostile_rows(driver, tmp_path):
    rows = np.array([[0.9, 0.1, 0.1, 0.5, 0.5],
                     [0.8, np.nan, 0.1, 0.5, 0.5],               # non-finite coordinates: no integer value
                     [0.7, -np.inf, 0.0, np.inf, 1.0],
                     [0.6, -3e30, -3e30, 3e30, 3e30],            # finite, far outside long long after scaling? (3e30 * 1920 < 9.2e18 is false)
                     [0.55, -1.0, -1.0, 2.0, 2.0],               # clamps to the whole frame
                     [0.5, 0.1, 0.1, 0.9, 0.9]], np.float32)     # not > 0.5
    rows[3, 1:] = [-4e15, -4e15, 4e15, 4e15]                      # * 1920 = 7.7e18 < 2^63: representable, then clamped
    p = tmp_path / "rows.bin"
    p.write_bytes(rows.tobytes())
    out = driver("rows", p, 1080, 1920, 0.5)
    assert out == "n=3 total=3 (192,108,768,432) (0,0,1920,1080) (0,0,1920,1080)", out
    rects = np.array([[10, 10, 50, 50], [11, 10, 50, 51], [12, 11, 49, 50], [300, 300, 40, 40], [10, 11, 51, 50]], np.int32)
    p = tmp_path / "rects.bin"
    p.write_bytes(rects.tobytes())
    assert driver("rects", p, 3) == "n=1 (11,10,50,50)"
    assert driver("rects", p, 0).startswith("n=5")
