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
