"""the reference's Haar-cascade fallback (face_detection.py:108-123).  A small cascade written by the
test in OpenCV's XML layout (bright blob on a darker surround: centre-surround, left/right and top/bottom balance
stumps over three stages) stands in for haarcascade_frontalface_default.xml, which ships with OpenCV, not with the
reference.  CPU: XML reader, the oracle's grouping against hand-worked cases.  GPU: dfd_detect_faces_haar equals the
oracle - candidate count and grouped boxes - on frames with planted blobs of several sizes, on noise and on a frame
smaller than the window; the host falls back to it when the handle has no SSD."""
import numpy as np
import pytest

from oracle import haar_ref

XML = """<?xml version="1.0"?>
<opencv_storage>
<cascade>
  <stageType>BOOST</stageType>
  <featureType>HAAR</featureType>
  <height>24</height>
  <width>24</width>
  <stageParams><boostType>GAB</boostType><maxWeakCount>3</maxWeakCount></stageParams>
  <featureParams><maxCatCount>0</maxCatCount><featSize>1</featSize><mode>BASIC</mode></featureParams>
  <stageNum>3</stageNum>
  <stages>
    <_><maxWeakCount>1</maxWeakCount><stageThreshold>0.5</stageThreshold>
      <weakClassifiers>
        <_><internalNodes>0 -1 0 0.7</internalNodes><leafValues>-1. 1.</leafValues></_>
      </weakClassifiers></_>
    <_><maxWeakCount>2</maxWeakCount><stageThreshold>1.5</stageThreshold>
      <weakClassifiers>
        <_><internalNodes>0 -1 1 -0.2</internalNodes><leafValues>-1. 1.</leafValues></_>
        <_><internalNodes>0 -1 1 0.2</internalNodes><leafValues>1. -1.</leafValues></_>
      </weakClassifiers></_>
    <_><maxWeakCount>3</maxWeakCount><stageThreshold>2.5</stageThreshold>
      <weakClassifiers>
        <_><internalNodes>0 -1 2 -0.2</internalNodes><leafValues>-1. 1.</leafValues></_>
        <_><internalNodes>0 -1 2 0.2</internalNodes><leafValues>1. -1.</leafValues></_>
        <_><internalNodes>0 -1 3 0.4</internalNodes><leafValues>-1. 1.</leafValues></_>
      </weakClassifiers></_>
  </stages>
  <features>
    <_><rects><_>0 0 24 24 -1.</_><_>6 6 12 12 4.</_></rects><tilted>0</tilted></_>
    <_><rects><_>0 0 24 24 -1.</_><_>0 0 12 24 2.</_></rects><tilted>0</tilted></_>
    <_><rects><_>0 0 24 24 -1.</_><_>0 0 24 12 2.</_></rects><tilted>0</tilted></_>
    <_><rects><_>2 2 20 20 -1.</_><_>8 8 8 8 6.25</_><_>0 0 1 1 0.5</_></rects><tilted>0</tilted></_>
  </features>
</cascade>
</opencv_storage>
"""


def _frame(h, w, blobs, seed):
    rs = np.random.RandomState(seed)
    f = rs.randint(30, 70, (h, w, 3)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for cx, cy, r in blobs:
        f += 150.0 * np.exp(-(((xx - cx) / r) ** 2 + ((yy - cy) / r) ** 2))[..., None]
    return np.clip(f, 0, 255).astype(np.uint8)


def test_xml_reader(pkg):
    cas = pkg.haar.load_cascade_xml(XML)
    assert cas["haar.win"].tolist() == [24.0, 24.0]
    assert cas["haar.stages"].tolist() == [[0, 1, 0.5], [1, 2, 1.5], [3, 3, 2.5]]
    assert cas["haar.stumps"].shape == (6, 4) and cas["haar.stumps"][2].tolist() == pytest.approx([1, 0.2, 1.0, -1.0])
    assert cas["haar.rects"].shape == (4, 15)
    assert cas["haar.rects"][0].tolist() == [0, 0, 24, 24, -1, 6, 6, 12, 12, 4, 0, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        pkg.haar.load_cascade_xml(XML.replace("<tilted>0</tilted>", "<tilted>1</tilted>", 1))
    with pytest.raises(ValueError):
        pkg.haar.load_cascade_xml(XML.replace("HAAR", "LBP"))


def test_group_rectangles_hand_cases():
    g = haar_ref.group_rectangles
    near = [(100 + d, 50 + d, 40, 40) for d in (0, 1, 2, 1, 0, 2, 1)]              # 7 similar rectangles -> one mean
    assert g(near, 5) == [(101, 51, 40, 40)]
    assert g(near[:5], 5) == []                                                   # needs MORE than minNeighbors members
    far = near + [(300, 200, 60, 60)] * 6
    assert g(far, 5) == [(101, 51, 40, 40), (300, 200, 60, 60)]                   # classes in order of first appearance
    inner = [(100, 100, 80, 80)] * 9 + [(120, 120, 30, 30)] * 6                   # a weaker small box inside a stronger big one
    assert g(inner, 5) == [(100, 100, 80, 80)]
    assert g([], 5) == [] and g(near, 0) == near


@pytest.fixture(scope="module")
def haar_handle(pkg, seeded_sd):
    cas = pkg.haar.load_cascade_xml(XML)
    h = pkg._lib.Handle(pkg.weights.pack_all(seeded_sd, None, None, haar=cas), device=0, max_batch=2)
    yield h, cas
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size,blobs,seed", [
    ((240, 320), [(80, 70, 14), (220, 150, 22)], 1),
    ((480, 640), [(150, 120, 20), (400, 300, 45), (560, 90, 11)], 2),
    ((1080, 1920), [(300, 300, 40), (1200, 600, 90), (1700, 200, 25)], 3),
    ((200, 260), [], 4),                                                            # noise only
    ((20, 300), [(100, 10, 8)], 5),                                                 # lower than the window: no scale fits
])
def test_hip_haar_equals_oracle(haar_handle, size, blobs, seed):
    h, cas = haar_handle
    frame = _frame(size[0], size[1], blobs, seed)
    want, ncand = haar_ref.detect(frame, cas)
    got, gcand = h.detect_faces_haar(frame, 1.1, 5, 30, with_candidates=True)
    assert gcand == ncand
    assert got == [tuple(int(v) for v in b) for b in want]
    if blobs and size[0] >= 100:
        assert len(got) >= 1, "no planted blob was found: the fixture does not exercise the detector"
    # other parameters take the same path
    want2, _ = haar_ref.detect(frame, cas, scale_factor=1.25, min_neighbors=2, min_size=40)
    assert h.detect_faces_haar(frame, 1.25, 2, 40) == [tuple(int(v) for v in b) for b in want2]


@pytest.mark.gpu
def test_host_falls_back_to_haar_without_ssd(pkg, haar_handle):
    h, cas = haar_handle
    assert h.has_haar and not h.has_detector
    frame = _frame(480, 640, [(150, 120, 20), (400, 300, 45)], 7)
    want, _ = haar_ref.detect(frame, cas)
    assert pkg.face_detection.detect_bounding_box(frame, handle=h) == [tuple(int(v) for v in b) for b in want]
    assert pkg.face_detection.detect_bounding_box(np.zeros((10, 10, 3), np.uint8), handle=h) == []


def _haar_stream():
    """planted blobs of several sizes, a noise-only frame (no detection -> 'frame_only'), a tiny frame"""
    return [_frame(480, 640, [(150, 120, 20), (400, 300, 45)], 11), _frame(480, 640, [(320, 240, 60)], 12),
            _frame(200, 260, [], 13), _frame(480, 640, [(500, 100, 25), (120, 350, 50), (330, 220, 18)], 14),
            _frame(720, 1280, [(640, 360, 70)], 15), _frame(480, 640, [(150, 120, 20), (400, 300, 45)], 16),
            _frame(20, 300, [(100, 10, 8)], 17), _frame(480, 640, [(320, 240, 60)], 18),
            _frame(480, 640, [(200, 200, 30)], 19), _frame(480, 640, [(420, 260, 40)], 20),
            _frame(480, 640, [(100, 380, 33)], 21), _frame(480, 640, [(320, 240, 60)], 22)]


@pytest.mark.gpu
def test_server_flow_on_a_haar_only_handle_matches_oracle(pkg, haar_handle, seeded_sd):
    """The reference AS SHIPPED has no SSD files, so `/analyze` detects with the cascade and classifies faces[0]
    (backend_server.py:153-171 -> face_detection.py:58-61,108-123).  A Haar-only handle must do exactly that inside
    the fused call: box == haar_ref.detect's first, 'face+frame', votes == the oracle flow driven by that detector."""
    from oracle.pipeline_ref import PredictRef

    h, cas = haar_handle
    det = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=h)
    ref = PredictRef(pkg.weights.to_torch(seeded_sd), None, None, detection_threshold=0.55, haar_cascade=cas)
    modes = []
    for frame in _haar_stream():
        want = ref.request(frame)
        got = det.analyze_request(frame)
        for k in ('analysis_mode', 'faces_detected', 'confidence_level', 'frame_count'):
            assert got[k] == want[k], k
        assert got.get('face_bbox') == want.get('face_bbox')
        assert abs(got['fake_probability'] - want['fake_probability']) <= 1e-3
        assert got['frame_forensic_probability'] == want['frame_forensic_probability']
        assert det.temporal_tracker.get_voting_stats() == want['votes']
        modes.append(got['analysis_mode'])
    assert modes.count('face+frame') >= 8 and modes.count('frame_only') >= 2, modes


@pytest.mark.gpu
def test_predict_on_a_haar_only_handle_matches_oracle(pkg, haar_handle, seeded_sd):
    """`predict` (deepfake_detection.py:603-626) classifies and votes on EVERY box the cascade returns."""
    from oracle.pipeline_ref import PredictRef

    h, cas = haar_handle
    det = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.5, handle=h)
    ref = PredictRef(pkg.weights.to_torch(seeded_sd), None, None, detection_threshold=0.5, haar_cascade=cas)
    total = 0
    for frame in _haar_stream()[:6]:
        want = ref.predict(frame)
        got = det.predict(frame)[3]
        assert got['faces_detected'] == want['faces_detected'] and got['analysis_mode'] == want['analysis_mode']
        assert [r['bbox'] for r in got['face_results']] == [r['bbox'] for r in want['face_results']]
        for g, w in zip(got['face_results'], want['face_results']):
            assert abs(g['face_prob'] - w['face_prob']) <= 1e-3
        assert det.temporal_tracker.get_voting_stats() == want['votes']
        assert got['confidence_level'] == want['confidence_level']
        total += got['faces_detected']
    assert total >= 6


@pytest.mark.gpu
def test_jpeg_request_and_health_on_a_haar_only_handle(pkg, haar_handle):
    """dfd_analyze_jpeg takes the same fallback (the decoded frame never visits the host), and /health reports face
    detection as available (reference backend_server.py:93: the DNN or its Haar fallback)."""
    import io

    from PIL import Image

    h, cas = haar_handle
    frame = _frame(480, 640, [(320, 240, 60)], 12)
    buf = io.BytesIO()
    Image.fromarray(frame[..., ::-1]).save(buf, "JPEG", quality=92)
    a = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=h)
    b = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=h)
    b.frame_analyzer.stream_id = a.frame_analyzer.stream_id + 1
    got = a.analyze_request(jpeg=buf.getvalue())
    decoded = np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))[..., ::-1])
    want = b.analyze_request(decoded)
    assert got == want and got['analysis_mode'] == 'face+frame' and got['faces_detected'] >= 1
    prev = pkg.runtime.peek_default_handle()
    pkg.runtime.set_default_handle(h)
    try:
        from rtdfd_amd import backend_server

        caps = backend_server.app.test_client().get('/health').get_json()['capabilities']
        assert caps['face_detection'] is True
    finally:
        pkg.runtime.set_default_handle(prev)
