"""SURVEY 8(f) N4, second half: test-time augmentation (reference deepfake_detection.py:408-443).  The augmented copies
(flip / convertScaleAbs / warpAffine) are bit-identical to the oracle's restatement of the three cv2 calls; the averaged
probability of DeepfakeDetector.analyze_face(use_tta=True) equals the oracle flow driven by the same `random` seed."""
import random

import numpy as np
import pytest
import torch

import frames as F
from oracle import b0_ref, imgproc_ref, tta_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,flip,alpha,angle", [((120, 96), True, 1.07, 2.4), ((224, 224), False, 0.91, -3.0),
                                                    ((81, 133), True, 1.1, 0.0), ((300, 260), False, 1.0, 1.3), ((40, 40), True, 0.9, -0.7)])
def test_augmented_copy_is_bit_identical(b0_handle, shape, flip, alpha, angle):
    img = F.natural_like(shape[0], shape[1], seed=shape[0] + shape[1])
    img[:6, :6] = 255                                               # saturating pixels for convertScaleAbs
    want = tta_ref.augment(img, flip, alpha, angle)
    got = b0_handle.tta_augment(img, flip, alpha, angle)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert not np.array_equal(got, img)


def test_tta_average_matches_oracle_flow(pkg, b0_handle, seeded_sd):
    det = pkg.deepfake_detection.DeepfakeDetector(use_tta=True, num_tta_augmentations=3, handle=b0_handle)
    face = F.natural_like(150, 130, seed=77)
    sd = pkg.weights.to_torch(seeded_sd)

    def single(img):
        x = torch.from_numpy(imgproc_ref.crop_resize_normalize(img)).unsqueeze(0)
        return float(torch.sigmoid(b0_ref.forward(sd, x).squeeze()).item())

    random.seed(1234)
    pre = imgproc_ref.preprocess_face_quality(face)
    preds = [single(pre)]
    for _ in range(2):
        flip = random.random() > 0.5
        br = random.uniform(0.9, 1.1)
        ang = random.uniform(-3, 3)
        preds.append(single(tta_ref.augment(pre, flip, br, ang)))
    want = float(np.mean(preds))
    random.seed(1234)
    got = det.analyze_face(face)
    assert got[2] is None and got[0] == got[1]
    assert abs(got[0] - want) <= 1e-3
    assert abs(got[0] - preds[0]) > 1e-6, "the augmented copies did not change the average"
    det1 = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, handle=b0_handle)
    assert abs(det1.analyze_face(face)[0] - preds[0]) <= 1e-3
