"""Regenerates tests/golden/*.json from the CPU oracle (`python tests/golden/make_golden.py`).

The reference itself cannot run here (cv2 / efficientnet_pytorch absent, weights missing), so these
vectors come from this repo's restatement with seeded weights and seeded inputs; they guard the
oracle against drift and give the GPU tests a fixed target that does not depend on torch-CPU's
kernels of the day.  Inputs are described by (generator, seed), never stored."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import frames as F  # noqa: E402
import rtdfd_amd  # noqa: E402
from oracle import b0_ref, mtcnn_ref, ssd_ref  # noqa: E402
from tests import mt_images  # noqa: E402
from oracle.forensics_ref import ForensicsRef  # noqa: E402


def b0_inputs(n=8, seed=42):
    rs = np.random.RandomState(seed)
    x = rs.randn(n, 3, 224, 224).astype(np.float32)
    return x * np.linspace(0.3, 2.0, n, dtype=np.float32).reshape(n, 1, 1, 1) + np.linspace(-1, 1, n, dtype=np.float32).reshape(n, 1, 1, 1)


FORENSIC_FRAMES = {"determinism": F.determinism_frame, "noisy": F.noisy_image, "gradient": F.gradient_image,
                   "smooth": F.smooth_image, "face_vga": F.face_frame, "natural_720p": F.natural_like}
SSD_FRAMES = {"face_vga": F.face_frame, "natural_720p": F.natural_like, "blank": F.blank_frame}


def main():
    W = rtdfd_amd.weights
    sd = W.to_torch(W.seeded_state_dict(0))
    x = b0_inputs()
    taps = {}
    logits = b0_ref.forward(sd, torch.from_numpy(x), taps)
    checks = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in taps.items() if k.endswith(".out") or k in ("stem", "head")}
    torch.manual_seed(42)
    ref_in = torch.randn(1, 3, 224, 224)                       # the reference's own determinism input
    json.dump({"weights_seed": 0, "inputs": "make_golden.b0_inputs(8, 42)", "logits": [float(v) for v in logits.flatten()],
               "block_checksums_sum_abssum": checks,
               "reference_determinism_input_logit": float(b0_ref.forward(sd, ref_in)[0, 0])},
              open(os.path.join(HERE, "b0_logits.json"), "w"), indent=1)

    out = {}
    for name, gen in FORENSIC_FRAMES.items():
        a = ForensicsRef()
        r = a.analyze(gen())
        out[name] = {"scores": r["scores"], "fake_probability": r["fake_probability"], "stats": a.stats}
    json.dump(out, open(os.path.join(HERE, "forensic_scores.json"), "w"), indent=1)

    ssd = W.to_torch(W.seeded_ssd_state_dict(0))
    S = rtdfd_amd.ssd_arch
    det = {}
    for name, gen in SSD_FRAMES.items():
        f = gen()
        rows = ssd_ref.forward(ssd, S, f)
        det[name] = {"n_rows": len(rows), "top_scores": [r[0] for r in rows[:5]],
                     "boxes": ssd_ref.postprocess(rows, f.shape[0], f.shape[1], 0.5)}
    json.dump({"weights_seed": 0, "frames": det}, open(os.path.join(HERE, "ssd_boxes.json"), "w"), indent=1)

    mt = W.to_torch(W.seeded_mtcnn_state_dict(0))
    cases = []
    for (h, w, seed), img in zip(mt_images.CASES, mt_images.images()):
        taps = {}
        face = mtcnn_ref.mtcnn_forward(mt, img, taps)
        cases.append({"image": f"mt_images.textured({h}, {w}, {seed})",
                      "rows": [len(taps[k]) for k in ("stage1", "stage2", "stage3")],
                      "selected": None if face is None else [float(v) for v in taps["selected"]],
                      "face_sum": None if face is None else int(face.sum()),
                      "face_corner": None if face is None else [int(v) for v in face[:, 0, :4].ravel()]})
    json.dump({"weights_seed": 0, "cases": cases}, open(os.path.join(HERE, "mtcnn_faces.json"), "w"), indent=1)
    print("wrote b0_logits.json forensic_scores.json ssd_boxes.json mtcnn_faces.json")


if __name__ == "__main__":
    main()
