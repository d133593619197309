"""GPU: bf16 activation storage (dfd_set_option(h, "bf16_activations", 1); BASELINE.json configs[3], SURVEY 8(d)
Config 4).  Every activation tensor that reaches HBM is bf16, all arithmetic and accumulation fp32, weights
fp32-exact (three bf16 planes) or - "bf16_weight_planes" = 1 - bf16.

bf16 is not held to the 1e-3 logit bar (that is the fp32 path's); the bars here are
  * every tap within 2 % of the tensor's absolute maximum of the fp32 oracle; logit error over 64 crops rms <= 1.6e-2,
    p95 <= 3.5e-2, max <= 1e-1 (a distribution, see test_bf16_logit_error_statistics),
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
    print(f"bf16 logit max|d| vs fp32 oracle (3 crops) = {err:.2e}; worst taps {sorted(worst.items(), key=lambda kv: -kv[1])[:3]}")
    assert err > 1e-6, "suspiciously exact: is the bf16 path running?"


def test_bf16_late_block_launches_stay_within_the_bf16_bars(pkg, bf16, seeded_sd):
    """Option "fuse_late" with bf16 activations (three products per K-step against the exact weight planes): the taps of
    the blocks it changes stay within the same relative bar as the default path."""
    x = _crops(3, 21)
    taps = {}
    b0_ref.forward(pkg.weights.to_torch(seeded_sd), torch.from_numpy(x), taps)
    xd = bf16.alloc(x.nbytes).upload(x)
    bf16.set_option("fuse_late", 1)
    try:
        for i in (6, 8, 9, 12, 15):
            for kind in ("dw", "out"):
                w = _nhwc(taps[f"b{i}.{kind}"]).reshape(-1)
                got = bf16.tap(xd.ptr, 3, f"b{i}.{kind}", w.size)
                rel = float(np.abs(got - w).max() / max(1e-6, np.abs(w).max()))
                assert rel <= 2e-2, f"b{i}.{kind}: {rel:.3e} of max|ref|"
    finally:
        bf16.set_option("fuse_late", 1)                    # the default
        xd.free()


def test_bf16_logit_error_statistics(pkg, bf16, seeded_sd):
    """The logit bar of the bf16 config, stated as what it is: a distribution.  Rounding every stored activation to
    bf16 (8 significand bits, 81 conv layers) makes the logit error a random variable - measured over 64 crops x 4
    kernel configurations x 3 seeds (profiles/bf16_logit_stats.py, round 3): rms 0.9e-2 .. 1.3e-2 (3 % of the logits'
    standard deviation of 0.3 .. 0.4), p95 1.5e-2 .. 2.9e-2, max 2.9e-2 .. 7.0e-2 - independent of whether the stem /
    expand convs are fused.  Any re-ordering of fp32 arithmetic upstream (1e-7 relative) flips bf16 roundings
    downstream and re-draws the sample: that is what moved the old 3-crop maximum from 2.6e-2 to 3.5e-2 when the stem
    conv went to the (fp32-exact, split-precision) MFMA in round 2 - the stem is NOT reduced in precision in this
    config, and the rms did not move (1.1e-2 before and after).  Bars: rms <= 1.6e-2, p95 <= 3.5e-2, max <= 1e-1."""
    x = _crops(64, 21)
    want = b0_ref.forward(pkg.weights.to_torch(seeded_sd), torch.from_numpy(x)).numpy().ravel()
    got = np.concatenate([bf16.classify(x[i:i + 16]) for i in range(0, 64, 16)]).ravel()      # handle capacity 16
    d = np.abs(got - want)
    rms, p95, mx = float(np.sqrt((d ** 2).mean())), float(np.quantile(d, 0.95)), float(d.max())
    print(f"bf16 logit error over 64 crops: rms {rms:.2e} p95 {p95:.2e} max {mx:.2e}; logit std {want.std():.3f}")
    assert rms <= 1.6e-2 and p95 <= 3.5e-2 and mx <= 1e-1
    assert rms > 1e-4, "suspiciously exact: is the bf16 path running?"


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


@pytest.fixture(scope="module")
def gate_data(pkg, b0_handle, seeded_sd, ssd_sd):
    """The 200-frame seeded stream through the server flow three ways: fp32 CPU ORACLE, fp32 HIP, bf16 HIP.  The
    per-request probability does not depend on the vote threshold, so one pass serves every threshold below."""
    h = b0_handle
    frames = _gate_stream(200)
    ref = PredictRef(pkg.weights.to_torch(seeded_sd), pkg.weights.to_torch(ssd_sd), pkg.ssd_arch, detection_threshold=0.5)
    ref_out = [ref.request(f) for f in frames]

    def run(mode):
        h.set_option("bf16_activations", mode)
        det = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.5, handle=h)
        return [det.analyze_request(f)['fake_probability'] for f in frames]

    try:
        p32, p16 = run(0), run(1)
    finally:
        h.set_option("bf16_activations", 0)
    return ref_out, p32, p16


def _votes(pkg, probs, thr):
    tr = pkg.tracker.TemporalTracker(voting_window=10, detection_threshold=thr)
    seq = []
    for p in probs:
        tr.update(p)
        st = tr.get_voting_stats()
        seq.append((tr.get_confidence_level(), st['fake_count'], st['real_count']))
    return seq


def test_config4_gate_votes_equal_fp32_oracle_on_200_frames(pkg, gate_data):
    """SURVEY 8(d) Config 4: votes / verdicts of the bf16 pipeline == the fp32 oracle's on a 200-frame seeded stream
    (server flow: faces[0], threshold placed INSIDE the stream's probability distribution so that both votes occur).
    This case proves the plumbing at a comfortable threshold; `test_config4_gate_threshold_sweep` is the robustness
    statement."""
    ref_out, p32, p16 = gate_data
    ref_probs = [r['fake_probability'] for r in ref_out]
    face = sorted(r['fake_probability'] for r in ref_out if r['analysis_mode'] == 'face+frame' and r['fake_probability'] < 0.99)
    lo, hi = len(face) // 4, 3 * len(face) // 4
    gaps = [(face[i + 1] - face[i], (face[i + 1] + face[i]) / 2) for i in range(lo, hi)]
    half_gap, thr = max(gaps)[0] / 2, max(gaps)[1]                  # the widest gap in the central half
    want = _votes(pkg, ref_probs, thr)
    err32 = max(abs(a - b) for a, b in zip(p32, ref_probs))
    err16 = max(abs(a - b) for a, b in zip(p16, ref_probs))
    near = [abs(a - b) for a, b in zip(p16, ref_probs) if abs(b - thr) < 0.05]
    at_risk = sum(1 for p in ref_probs if abs(p - thr) <= max(near))         # frames a bf16-sized error could flip
    fakes = sum(1 for p in ref_probs if p > thr)
    print(f"gate: thr={thr:.5f} half-gap={half_gap:.2e} fake votes {fakes}/200; max|dp| fp32 {err32:.2e}, bf16 {err16:.2e} "
          f"(logit scale: dp / p(1-p) ~ {err16 / 0.18:.2e}); frames within the bf16 error of the threshold: {at_risk}")
    assert 40 <= fakes <= 160, "threshold does not split the stream"
    assert err32 <= 2e-3 and _votes(pkg, p32, thr) == want          # the fp32 HIP path
    assert err16 <= BF16_PROB_BOUND
    print(f"gate: bf16 error of the frames within 0.05 of the threshold: max {max(near):.2e} over {len(near)} frames")
    got = _votes(pkg, p16, thr)
    assert got == want, "bf16 votes / verdicts differ from the fp32 oracle"
    assert {lv for lv, _, _ in want} >= {'FAKE', 'REAL'}


# bound on |p_bf16 - p_fp32 oracle| per request used to classify a threshold as knife-edge: measured max 6.0e-3 on this
# stream (round 2), 8e-3 leaves the margin the vote test above asserts
BF16_PROB_BOUND = 8e-3


def test_config4_gate_threshold_sweep(pkg, gate_data):
    """what the bf16 path does to the vote at thresholds nobody picked for it - the reference's own
    0.5 (`DeepfakeDetector`, deepfake_detection.py:730) and 0.55 (backend_server.py:57) and 32 thresholds spread over
    the quantiles of the stream's probabilities.  Per threshold: flipped votes (frames whose FAKE/REAL vote differs
    from the fp32 ORACLE's) and flipped verdicts (frames at which the 10-vote majority differs).  A threshold is
    `knife-edge` when some oracle probability lies within BF16_PROB_BOUND of it - there a bf16-sized error may
    legitimately move a vote; everywhere else the sequences must be IDENTICAL.  On knife-edge thresholds the flips are
    bounded by the frames at risk and listed in the output."""
    ref_out, p32, p16 = gate_data
    ref_probs = np.array([r['fake_probability'] for r in ref_out])
    qs = np.quantile(ref_probs, np.linspace(0.03, 0.97, 32))
    thresholds = [0.5, 0.55] + [float(q) for q in qs]
    rows, clean, total_flips16, total_flips32 = [], 0, 0, 0
    for thr in thresholds:
        want = _votes(pkg, ref_probs, thr)
        got16, got32 = _votes(pkg, p16, thr), _votes(pkg, p32, thr)
        vote_flips16 = int(sum((a > thr) != (b > thr) for a, b in zip(p16, ref_probs)))
        vote_flips32 = int(sum((a > thr) != (b > thr) for a, b in zip(p32, ref_probs)))
        verdict_flips16 = sum(1 for a, b in zip(got16, want) if a[0] != b[0])
        at_risk = int(np.sum(np.abs(ref_probs - thr) <= BF16_PROB_BOUND))
        rows.append((thr, at_risk, vote_flips16, verdict_flips16, vote_flips32))
        total_flips16 += vote_flips16
        total_flips32 += vote_flips32
        if at_risk == 0:
            clean += 1
            assert got16 == want, f"threshold {thr:.5f}: no oracle probability within {BF16_PROB_BOUND} of it, yet bf16 votes differ"
        else:
            assert vote_flips16 <= at_risk, f"threshold {thr:.5f}: {vote_flips16} flipped votes, only {at_risk} frames at risk"
        assert vote_flips32 <= int(np.sum(np.abs(ref_probs - thr) <= 2e-3)), f"fp32 HIP path flips votes at {thr:.5f}"
    print("gate sweep: threshold  frames_within_bound  bf16_flipped_votes  bf16_flipped_verdict_frames  fp32hip_flipped_votes")
    for r in rows:
        print(f"gate sweep: {r[0]:.5f}  {r[1]:3d}  {r[2]:3d}  {r[3]:3d}  {r[4]:3d}" + ("   [knife-edge]" if r[1] else ""))
    print(f"gate sweep: {clean} of {len(thresholds)} thresholds have no frame within {BF16_PROB_BOUND} (identical votes asserted); "
          f"flipped votes over all thresholds: bf16 {total_flips16}, fp32 HIP {total_flips32} of {200 * len(thresholds)}")
    assert float(np.abs(np.array(p16) - ref_probs).max()) <= BF16_PROB_BOUND
