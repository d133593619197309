"""Multi-GPU vote exchange, rehearsed on CPU: world_size 2 and 3 over gloo.  Every rank must end
with the same tracker state as a single process that saw all frames in order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _probs(n):
    """per frame 0..2 'faces' with deterministic probabilities; some frames contribute no vote"""
    rs = np.random.RandomState(123)
    out = []
    for t in range(n):
        k = int(rs.randint(0, 3))
        out.append([float(p) for p in rs.rand(k)] if k else [None if t % 7 == 0 else float(rs.rand())])
    return out


def _worker(rank, world, port, n_frames, wave, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import rtdfd_amd
    from rtdfd_amd.distributed import ShardedVote, shard_frames
    from rtdfd_amd.tracker import TemporalTracker

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    probs = _probs(n_frames)
    mine = set(shard_frames(n_frames, rank, world))
    tr = TemporalTracker(voting_window=10, detection_threshold=0.5)
    sv = ShardedVote(tr, capacity=16)
    verdicts = []
    for start in range(0, n_frames, wave * world):            # a wave = `wave` frames per rank
        for t in range(start, min(start + wave * world, n_frames)):
            if t in mine:
                sv.add(t, probs[t])
        verdicts.append(sv.finish_wave())
    q.put((rank, verdicts, tr.get_voting_stats(), list(tr.score_history), tr.get_stability_score()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,wave", [(2, 1), (2, 4), (3, 2)])
def test_sharded_vote_equals_single_process(pkg, world, wave):
    n_frames = 41
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, wave, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process truth: all frames in order, verdict sampled at the same wave boundaries
    tr = pkg.tracker.TemporalTracker(voting_window=10, detection_threshold=0.5)
    probs = _probs(n_frames)
    truth = []
    for start in range(0, n_frames, wave * world):
        for t in range(start, min(start + wave * world, n_frames)):
            for p in probs[t]:
                tr.update(p)
        truth.append(tr.get_confidence_level())
    for rank, verdicts, stats, history, stab in results:
        assert verdicts == truth, rank
        assert stats == tr.get_voting_stats()
        assert history == list(tr.score_history)                 # bit-identical doubles on every rank
        assert stab == tr.get_stability_score()
    assert "FAKE" in truth or "REAL" in truth


def test_capacity_overflow_is_loud(pkg):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(ValueError):
            pkg.distributed.gather_records([(i, 0, 0.5) for i in range(5)], capacity=4)
        out = pkg.distributed.gather_records([(3, 1, 0.25), (3, 0, None), (1, 0, 0.75)], capacity=4)
        assert out == [(1, 0, 0.75), (3, 0, None), (3, 1, 0.25)]
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_path_single_rank(pkg):
    """backend "nccl" (= RCCL) with device tensors, world size 1 on the one-GPU box."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        tr = pkg.tracker.TemporalTracker()
        sv = pkg.distributed.ShardedVote(tr)
        for t in range(12):
            sv.add(t, [0.9])
            level = sv.finish_wave()
        assert level == "FAKE" and tr.get_voting_stats()["fake_count"] == 10
    finally:
        dist.destroy_process_group()
