"""GPU: the frame-sharded stream driver (streams.ShardedStreams) on the device path.
  * world 1: verdicts, probabilities and vote counts equal the stateful single-GPU /analyze flow
    (DeepfakeDetector.analyze_request: dfd_analyze_frame with per-stream temporal state in the handle);
  * two ranks emulated on the one GPU (each computes its own shard from frames t-1 and t; blocks concatenated the
    way the all-gather returns them): every rank's sequence equals the world-1 sequence - in particular the
    temporal forensic signal, whose gray(t-1) the owning rank recomputes;
  * the RCCL transport through the C ABI (dfd_comm_* + dfd_vote_allgather) at world size 1."""
import numpy as np
import pytest

import frames as F

pytestmark = pytest.mark.gpu

H, W = 480, 640


def _stream(seed, n):
    """slowly changing frames (temporal signal active), blank frames (no face -> forensic vote) mixed in"""
    rs = np.random.RandomState(seed)
    base = F.natural_like(H, W, seed=40 + seed).astype(np.int16)
    out = []
    for t in range(n):
        if t % 5 == 4:
            out.append(F.blank_frame(W, H))
            continue
        amp = (0, 1, 0, 4, 0, 9)[t % 6]
        out.append(np.clip(base + rs.randint(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8))
    return out


def _run_world(pkg, h, streams, world, transport="local"):
    S = pkg.streams
    n_streams, n_frames = len(streams), len(streams[0])
    ranks = [S.ShardedStreams(h, n_streams, r, world, transport=transport if world == 1 else "local") for r in range(world)]
    seq = [{s: [] for s in range(n_streams)} for _ in range(world)]
    for wave in range((n_frames + world - 1) // world):
        blocks = []
        for sh in ranks:
            t = sh.frame_of(wave)
            block = np.full((n_streams, S.RECORD_FLOATS), -1.0)
            if t < n_frames:
                items = [(s, t, t > 0) for s in range(n_streams)]
                batch = [streams[s][t] for s in range(n_streams)] + ([streams[s][t - 1] for s in range(n_streams)] if t > 0 else [])
                arr = np.stack(batch)
                fd = h.alloc(arr.nbytes).upload(arr)
                block = sh.local_records(fd.ptr, H, W, items, conf_thr=0.5)
                fd.free()
            blocks.append(block)
        gathered = np.concatenate(blocks)                      # rank-major, as ncclAllGather returns it
        for r, sh in enumerate(ranks):
            out = sh.finish_wave(blocks[0]) if world == 1 else S.replay_all(sh.replicas, gathered)
            for s, rows in out.items():
                seq[r][s] += rows
    return ranks, seq


def test_world1_equals_stateful_single_gpu_flow_and_two_ranks_equal_world1(pkg, b0_handle):
    h = b0_handle
    streams = [_stream(1, 14), _stream(2, 14)]
    ranks1, seq1 = _run_world(pkg, h, streams, 1)
    # the stateful per-frame API, one detector object per stream
    for s, frames in enumerate(streams):
        det = pkg.deepfake_detection.DeepfakeDetector(use_tta=False, num_tta_augmentations=1, detection_threshold=0.55, handle=h)
        det.frame_analyzer.stream_id = 700 + s
        h.forensics_reset(700 + s)
        for t, f in enumerate(frames):
            want = det.analyze_request(f)
            got = seq1[0][s][t]
            assert got['analysis_mode'] == want['analysis_mode'], (s, t)
            assert got['frame_forensic_probability'] == want['frame_forensic_probability'], (s, t)
            assert got['fake_probability'] == want['fake_probability'], (s, t)
            assert got['confidence_level'] == want['confidence_level'], (s, t)
        assert ranks1[0].replicas[s].tracker.get_voting_stats() == det.temporal_tracker.get_voting_stats()
    modes = {r['analysis_mode'] for s in seq1[0].values() for r in s}
    assert modes == {'face+frame', 'frame_only'}, "the stream exercises only one branch of the vote"
    assert any(r['confidence_level'] in ('FAKE', 'REAL') for r in seq1[0][0])
    # two ranks, frame t on rank t % 2
    ranks2, seq2 = _run_world(pkg, h, streams, 2)
    for r in range(2):
        for s in range(len(streams)):
            assert seq2[r][s] == seq1[0][s], (r, s)
            assert list(ranks2[r].replicas[s].diffs) == list(ranks1[0].replicas[s].diffs)


def test_rccl_transport_world1(pkg, b0_handle):
    h = b0_handle
    cid = h.comm_unique_id()
    assert len(cid) == 128
    h.comm_init(cid, 0, 1)
    try:
        assert h.comm_info() == (0, 1)
        block = np.arange(30, dtype=np.float64).reshape(3, 10)
        out = h.vote_allgather(block)
        assert out.shape == (1, 3, 10) and np.array_equal(out[0], block)
        streams = [_stream(5, 6)]
        _, a = _run_world(pkg, h, streams, 1, transport="rccl")
        _, b = _run_world(pkg, h, streams, 1, transport="local")
        assert a == b
        with pytest.raises(pkg._lib.DfdError):
            h.comm_init(cid, 0, 1)                              # one communicator per handle
    finally:
        h.comm_destroy()
    assert h.comm_info()[1] == 0
    with pytest.raises(pkg._lib.DfdError):
        h.vote_allgather(np.zeros((1, 10)))


def test_lookahead_batches_give_the_per_wave_records(pkg, b0_handle):
    """local_records_waves (L waves of a rank in one device pass) returns, wave by wave, exactly the blocks that L
    calls of local_records return - the look-ahead only changes how frames are batched on the device."""
    S = pkg.streams
    h = b0_handle
    streams = [_stream(3, 7), _stream(4, 7)]
    n_streams = len(streams)
    for world, rank in ((1, 0), (2, 1)):
        sh = S.ShardedStreams(h, n_streams, rank, world, transport="local")
        waves = [w for w in range(7) if sh.frame_of(w) < 7]
        single = []
        for w in waves:
            t = sh.frame_of(w)
            arr = np.stack([streams[s][t] for s in range(n_streams)] + ([streams[s][t - 1] for s in range(n_streams)] if t > 0 else []))
            fd = h.alloc(arr.nbytes).upload(arr)
            single.append(sh.local_records(fd.ptr, H, W, [(s, t, t > 0) for s in range(n_streams)]))
            fd.free()
        cur, prev, items = [], [], []
        for w in waves:
            t = sh.frame_of(w)
            cur += [streams[s][t] for s in range(n_streams)]
            if t > 0:
                prev += [streams[s][t - 1] for s in range(n_streams)]
            items.append([(s, t, t > 0) for s in range(n_streams)])
        arr = np.stack(cur + prev)
        fd = h.alloc(arr.nbytes).upload(arr)
        multi = sh.local_records_waves(fd.ptr, H, W, items)
        fd.free()
        assert len(multi) == len(single)
        for a, b in zip(single, multi):
            assert np.array_equal(a, b, equal_nan=True), (world, rank)


def test_predecessor_only_frames_skip_the_signals_not_the_differences(pkg, b0_handle):
    """prev_index = -2 marks frames that are only somebody's predecessor: the scored frames get exactly the values of
    a call that analyses every frame; -2 outside the tail of the batch is an argument error."""
    h = b0_handle
    cur = _stream(5, 3)
    prev = _stream(6, 2)
    arr = np.stack(cur + prev)
    fd = h.alloc(arr.nbytes).upload(arr)
    full_idx = np.array([3, 4, -1, -1, -1], np.int32)
    lean_idx = np.array([3, 4, -1, -2, -2], np.int32)
    s_full, d_full = h.forensic_signals_device(fd.ptr, 5, H, W, full_idx)
    s_lean, d_lean = h.forensic_signals_device(fd.ptr, 5, H, W, lean_idx)
    assert np.array_equal(s_full[:3], s_lean[:3]) and np.array_equal(d_full[:3], d_lean[:3])
    assert np.all(s_lean[3:] == -1.0) and np.all(d_lean[3:] == -1.0) and d_lean[0] >= 0 and d_lean[2] == -1.0
    with pytest.raises(pkg._lib.DfdError):
        h.forensic_signals_device(fd.ptr, 5, H, W, np.array([3, -2, -1, -1, -1], np.int32))
    fd.free()


def test_groups_on_two_handles_give_the_one_handle_records(pkg, b0_handle, seeded_sd):
    """`local_records_groups` on the device: three look-ahead groups computed two at a time (second handle: its own
    streams and workspaces, main stream from the high-priority pool, a second host thread) return the record blocks of the
    one-handle loop bit for bit - a record is a pure function of its frame and the predecessor, whichever handle runs it."""
    S = pkg.streams
    h = b0_handle
    streams = [_stream(5, 7), _stream(6, 7)]
    n_streams = len(streams)
    bufs, groups = [], []
    for t0 in (0, 2, 4):                                          # groups of two waves (frames t0, t0 + 1)
        cur, prev, items = [], [], []
        for t in (t0, t0 + 1):
            cur += [streams[s][t] for s in range(n_streams)]
            if t > 0:
                prev += [streams[s][t - 1] for s in range(n_streams)]
            items.append([(s, t, t > 0) for s in range(n_streams)])
        arr = np.stack(cur + prev)
        bufs.append(h.alloc(arr.nbytes).upload(arr))
        groups.append((bufs[-1].ptr, H, W, items))
    other = pkg._lib.Handle(pkg.weights.pack_all(seeded_sd, pkg.weights.seeded_ssd_state_dict(0)), device=0, max_batch=16)
    try:
        other.set_option("stream_priority", 1)
        one = S.ShardedStreams(h, n_streams).local_records_groups(groups)
        two = S.ShardedStreams(h, n_streams, extra_handles=[other]).local_records_groups(groups)
        assert len(one) == len(two) == 3
        for a, b in zip(one, two):
            assert len(a) == len(b) == 2
            for x, y in zip(a, b):
                assert np.array_equal(x, y, equal_nan=True)
        assert any(np.isfinite(blk[:, S.F_FACE_PROB]).any() for a in one for blk in a)      # faces were classified
    finally:
        for b in bufs:
            b.free()
        other.close()
