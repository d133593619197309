"""GPU: bf16 activation storage (dfd_set_option(h, "bf16_activations", 1); BASELINE.json configs[3], SURVEY 8(d)
Config 4).  Every activation tensor that reaches HBM is bf16, all arithmetic and accumulation fp32, weights
fp32-exact (three bf16 planes) or - "bf16_weight_planes" = 1 - bf16.

bf16 is not held to the 1e-3 logit bar (that is the fp32 path's); the bars here are
  * every tap within 2 % of the tensor's absolute maximum of the fp32 oracle, logits within 6e-2,
  * bit-identical results across GEMM tiles, batch sizes and runs,
  * the Config 4 gate: on a 200-frame seeded stream the votes and verdicts of the bf16 pipeline equal those of the
    fp32 ORACLE (CPU), with the logit error reported separately."""
import numpy as np
import pytest
import torch

import frames as F
from oracle import b0_ref
from oracle.pipeline_ref import PredictRef

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16(b0_handle):
    b0_handle.set_option("bf16_activations", 1)
    b0_handle.set_option("bf16_weight_planes", 3)
    yield b0_handle
    b0_handle.set_option("bf16_activations", 0)
    b0_handle.set_option("bf16_weight_planes", 3)


def _crops(n, seed):
    rs = np.random.RandomState(seed)
    return (rs.randn(n, 3, 224, 224) * np.linspace(0.4, 1.8, n).reshape(n, 1, 1, 1)).astype(np.float32)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy() if t.dim() == 4 else t.numpy()


@pytest.mark.parametrize("fuse", [0, 1])
def test_bf16_taps_and_logits_close_to_fp32_oracle(pkg, bf16, seeded_sd, fuse):
    x = _crops(3, 21)
    taps = {}
    want = b0_ref.forward(pkg.weights.to_torch(seeded_sd), torch.from_numpy(x), taps).numpy()
    xd = bf16.alloc(x.nbytes).upload(x)
    bf16.set_option("fuse_expand", fuse)
    bf16.set_option("fuse_stem", fuse)
    names = ["stem"] + [f"b{i}.{k}" for i in range(16) for k in ("dw", "gate", "out")] + ["head"]
    if not fuse:
        names += [f"b{i}.exp" for i in range(1, 16)]
    worst = {}
    try:
        for name in names:
            w = _nhwc(taps[name]).reshape(-1)
            got = bf16.tap(xd.ptr, 3, name, w.size)
            rel = float(np.abs(got - w).max() / max(1e-6, np.abs(w).max()))
            worst[name] = rel
            assert rel <= 2e-2, f"{name}: {rel:.3e} of max|ref|"
        got = bf16.classify(x)
    finally:
        bf16.set_option("fuse_expand", 1)
        bf16.set_option("fuse_stem", 1)
        xd.free()
    err = float(np.abs(got - want).max())
    print(f"bf16 logit max|d| vs fp32 oracle = {err:.2e}; worst taps {sorted(worst.items(), key=lambda kv: -kv[1])[:3]}")
    assert err <= 6e-2          # bf16 keeps ~3 significant digits per stored activation; observed 6e-3 .. 3.5e-2
    assert err > 1e-6, "suspiciously exact: is the bf16 path running?"


@pytest.mark.parametrize("planes", [3, 1])
def test_bf16_tiles_batches_and_runs_are_bit_identical(pkg, bf16, planes):
    bf16.set_option("bf16_weight_planes", planes)
    x = _crops(7, 5)
    full = bf16.classify(x)
    assert np.array_equal(bf16.classify(x), full)
    for i in (0, 3, 6):
        assert np.array_equal(bf16.classify(x[i:i + 1]), full[i:i + 1])
    assert np.array_equal(bf16.classify(x[2:6]), full[2:6])
    try:
        for i in range(pkg._lib.load().dfd_gemm_tile_count()):
            bf16.set_option("gemm_tile", i)
            assert np.array_equal(bf16.classify(x[:3]), full[:3]), f"tile {i}"
    finally:
        bf16.set_option("gemm_tile", -1)


def _gate_stream(n):
    out = []
    for i in range(n):
        if i % 9 == 8:
            out.append(F.blank_frame(640, 480))                     # no face: the forensic probability is voted
        else:
            out.append(F.varied_frame(i))
    return out


def test_config4_gate_votes_equal_fp32_oracle_on_200_frames(pkg, b0_handle, seeded_sd, ssd_sd):
    """SURVEY 8(d) Config 4: votes / verdicts of the bf16 pipeline == the fp32 oracle's on a 200-frame seeded stream
    (server flow: faces[0], threshold placed INSIDE the stream's probability distribution so that both votes occur)."""
    h = b0_handle
    frames = _gate_stream(200)
    ref = PredictRef(pkg.weights.to_torch(seeded_sd), pkg.weights.to_torch(ssd_sd), pkg.ssd_arch, detection_threshold=0.5)
    ref_out = [ref.request(f) for f in frames]
    ref_probs = [r['fake_probability'] for r in ref_out]
    face = sorted(r['fake_probability'] for r in ref_out if r['analysis_mode'] == 'face+frame' and r['fake_probability'] < 0.99)
    lo, hi = len(face) // 4, 3 * len(face) // 4
    gaps = [(face[i + 1] - face[i], (face[i + 1] + face[i]) / 2) for i in range(lo, hi)]
    half_gap, thr = max(gaps)[0] / 2, max(gaps)[1]                  # the widest gap in the central half

    def votes(probs):
        tr = pkg.tracker.TemporalTracker(voting_window=10, detection_threshold=thr)
        seq = []
        for p in probs:
            tr.update(p)
            seq.append((tr.get_confidence_level(), tr.get_voting_stats()['fake_count'], tr.get_voting_stats()['real_count']))
        return seq

    def run(mode):
        h.set_option("bf16_activations", mode)
        det = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=thr, handle=h)
        return [det.analyze_request(f)['fake_probability'] for f in frames]

    try:
        p32, p16 = run(0), run(1)
    finally:
        h.set_option("bf16_activations", 0)
    want = votes(ref_probs)
    err32 = max(abs(a - b) for a, b in zip(p32, ref_probs))
    err16 = max(abs(a - b) for a, b in zip(p16, ref_probs))
    near = [abs(a - b) for a, b in zip(p16, ref_probs) if abs(b - thr) < 0.05]
    at_risk = sum(1 for p in ref_probs if abs(p - thr) <= max(near))         # frames a bf16-sized error could flip
    fakes = sum(1 for p in ref_probs if p > thr)
    print(f"gate: thr={thr:.5f} half-gap={half_gap:.2e} fake votes {fakes}/200; max|dp| fp32 {err32:.2e}, bf16 {err16:.2e} "
          f"(logit scale: dp / p(1-p) ~ {err16 / 0.18:.2e}); frames within the bf16 error of the threshold: {at_risk}")
    assert 40 <= fakes <= 160, "threshold does not split the stream"
    assert err32 <= 2e-3 and votes(p32) == want                    # the fp32 HIP path
    assert err16 <= 8e-3
    print(f"gate: bf16 error of the frames within 0.05 of the threshold: max {max(near):.2e} over {len(near)} frames")
    got = votes(p16)
    assert got == want, "bf16 votes / verdicts differ from the fp32 oracle"
    assert {lv for lv, _, _ in want} >= {'FAKE', 'REAL'}
