"""Frame-sharded streams: frame t of a stream on rank t % G, ONE all-gather of fixed-size records per wave,
then every rank replays temporal forensic score, weighted sum and vote in frame order (BASELINE.json
configs[4]).

What crosses frames in the reference's /analyze flow (backend_server.py:147-233) is
  * the analyzer's temporal signal: previous gray frame + the last 30 mean differences + its frame counter
    (frame_analysis.py:349-389) and the full/fast schedule on the detector's frame counter
    (deepfake_detection.py:509-512),
  * the TemporalTracker (deepfake_detection.py:93-290).
Everything else is a pure function of one frame.  A rank therefore computes per frame: the face
probability of faces[0] (or none), the five stateless forensic scores and mean|gray(t) - gray(t-1)| - it holds
frame t-1 too and recomputes its gray plane (64 KB) instead of receiving it - and ships them as one
80-byte record.  `StreamReplica.replay` is the rest of the flow; fed the records of all ranks in frame order
it performs exactly the arithmetic of the single-GPU stateful path (csrc/forensic_api.hip forensics_run +
DeepfakeDetector.analyze_request), so every rank ends each wave with bit-identical vote state.

Transports for the exchange: "rccl" = dfd_vote_allgather through the C ABI (ncclAllGather on the handle's
stream), "torch" = torch.distributed.all_gather_into_tensor (gloo in the CPU tests), "local" = world size 1.
"""
from __future__ import annotations

import math
from collections import deque
from typing import Dict, List, Optional, Sequence

import numpy as np

from .tracker import TemporalTracker

# record layout (float64 each)
F_STREAM, F_FRAME, F_FACE_PROB, F_MEAN_DIFF, F_FREQ, F_NOISE, F_ELA, F_EDGE, F_COLOR, F_NFACES = range(10)
RECORD_FLOATS = 10
FULL_WEIGHTS = (0.25, 0.20, 0.20, 0.15, 0.10, 0.10)      # frequency, noise, ela, edge, color, temporal (:49-56)


def _clip01(v: float) -> float:
    return 0.0 if v < 0.0 else (1.0 if v > 1.0 else v)


def _sigmoid32(logit) -> float:
    x = np.float32(logit)
    return float(np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32)))


class StreamReplica:
    """All cross-frame state of one /analyze stream; identical on every rank after each replayed record."""

    def __init__(self, detection_threshold: float = 0.55, full_forensic_interval: int = 3):
        self.tracker = TemporalTracker(window_size=60, high_confidence_threshold=0.6, voting_window=10,
                                       detection_threshold=detection_threshold)
        self.full_forensic_interval = full_forensic_interval
        self.diffs: deque = deque(maxlen=30)          # frame_analysis.py:36
        self.analyzer_frames = 0                      # FrameForensicAnalyzer.frame_count
        self.frame_count = 0                          # DeepfakeDetector.frame_count
        self.next_frame = 0

    def forensic_probability(self, rec) -> float:
        """temporal score + weighted sum for this frame (the host half of csrc/forensic_api.hip forensics_run)"""
        full = self.frame_count % self.full_forensic_interval == 0     # before the counter moves (server order)
        self.analyzer_frames += 1
        temporal = 0.0
        md = float(rec[F_MEAN_DIFF])
        if md >= 0.0:                                                  # a predecessor exists
            self.diffs.append(md)
            if len(self.diffs) >= 5:
                d = list(self.diffs)
                m = 0.0
                for v in d:
                    m += v
                m /= len(d)
                q = 0.0
                for v in d:
                    q += (v - m) * (v - m)
                cv = math.sqrt(q / len(d)) / (m + 1e-10)
                s = 0.0
                if cv > 1.5:
                    s += 0.4
                elif cv > 1.0:
                    s += 0.2
                if md < 0.3 and self.analyzer_frames > 10:
                    s += 0.3
                elif md < 0.8 and self.analyzer_frames > 10:
                    s += 0.1
                temporal = _clip01(s)
        comb = 0.0
        if full:
            sc = (rec[F_FREQ], rec[F_NOISE], rec[F_ELA], rec[F_EDGE], rec[F_COLOR], temporal)
            for v, w in zip(sc, FULL_WEIGHTS):
                comb += float(v) * w
        else:
            comb += float(rec[F_FREQ]) * 0.45
            comb += temporal * 0.25
            comb += float(rec[F_EDGE]) * 0.30
        return _clip01(comb)

    def replay(self, rec) -> dict:
        frame = int(rec[F_FRAME])
        if frame != self.next_frame:
            raise ValueError(f"records must arrive in frame order: got frame {frame}, expected {self.next_frame}")
        self.next_frame += 1
        fprob = self.forensic_probability(rec)
        self.frame_count += 1
        p = rec[F_FACE_PROB]
        vote = fprob if np.isnan(p) else float(p)                     # backend_server.py:166,199
        self.tracker.update(vote)
        return {'frame': frame, 'analysis_mode': 'frame_only' if np.isnan(p) else 'face+frame',
                'fake_probability': vote, 'frame_forensic_probability': fprob,
                'confidence_level': self.tracker.get_confidence_level(), 'faces_detected': int(rec[F_NFACES])}


def exchange(block: np.ndarray, transport: str = "local", handle=None, group=None) -> np.ndarray:
    """(capacity, 10) float64 of this rank -> (world * capacity, 10) of all ranks, rank-major."""
    block = np.ascontiguousarray(block, np.float64)
    if transport == "local":
        return block
    if transport == "rccl":
        return handle.vote_allgather(block).reshape(-1, RECORD_FLOATS)
    if transport == "torch":
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        send = torch.from_numpy(block).to(dev)
        recv = torch.empty((world * block.shape[0], RECORD_FLOATS), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        return recv.cpu().numpy()
    raise ValueError(f"unknown transport {transport!r}")


def exchange_waves(blocks: Sequence[np.ndarray], transport: str = "local", handle=None, group=None) -> List[np.ndarray]:
    """The record blocks of consecutive waves -> per wave the (world * capacity, 10) records of all ranks.  "rccl": ONE
    library call (dfd_vote_allgather_waves: one upload, an ncclAllGather per wave, one download, one wait); the other
    transports exchange wave by wave."""
    if transport == "rccl" and len(blocks) > 1:
        stacked = np.ascontiguousarray(np.stack(blocks), np.float64)
        out = handle.vote_allgather_waves(stacked)                    # (waves, world, capacity, 10)
        return [out[w].reshape(-1, RECORD_FLOATS) for w in range(len(blocks))]
    return [exchange(b, transport, handle, group) for b in blocks]


def replay_all(replicas: Sequence[StreamReplica], records: np.ndarray) -> Dict[int, List[dict]]:
    """Feed gathered records (unused slots have stream < 0) to the replicas in (frame, stream) order."""
    rows = [r for r in records if r[F_STREAM] >= 0]
    rows.sort(key=lambda r: (r[F_FRAME], r[F_STREAM]))
    out: Dict[int, List[dict]] = {}
    for r in rows:
        s = int(r[F_STREAM])
        out.setdefault(s, []).append(replicas[s].replay(r))
    return out


class ShardedStreams:
    """Per-rank driver.  A wave = one frame of every stream on every rank (frame index wave * G + rank)."""

    def __init__(self, handle, n_streams: int, rank: int = 0, world: int = 1, transport: Optional[str] = None,
                 detection_threshold: float = 0.55, group=None, calibrator=None, extra_handles: Sequence = ()):
        self.h, self.n_streams, self.rank, self.world, self.group = handle, n_streams, rank, world, group
        # further handles on the same device (each with its own streams and workspaces): `local_records_groups` runs
        # look-ahead groups on them side by side.  The exchange always goes through `handle` (it owns the communicator).
        self.workers = [handle, *extra_handles]
        self.transport = transport or ("local" if world == 1 else "torch")
        self.replicas = [StreamReplica(detection_threshold) for _ in range(n_streams)]
        # `DeepfakeDetector.calibrator` of the single-GPU flow (reference deepfake_detection.py:336-342,445-455): a
        # deployment with weights/calibrator.pkl must vote the same sharded and unsharded
        self.calibrator = calibrator

    def face_probability(self, logit, h: int, w: int) -> float:
        """sigmoid -> calibration -> +0.10 for crops under 80 px -> clip: `DeepfakeDetector._finish_face`
        (reference deepfake_detection.py:397-398,445-455,489-502)"""
        p = _sigmoid32(logit)
        if self.calibrator is not None:
            try:
                p = self.calibrator.predict_proba([[p]])[0][1]
            except Exception:                                       # reference :454-455
                pass
        return float(np.clip(p + (0.10 if (h < 80 or w < 80) else 0.0), 0, 1))

    def frame_of(self, wave: int) -> int:
        return wave * self.world + self.rank

    def local_records(self, frames_dev: int, height: int, width: int, items, conf_thr: float = 0.5) -> np.ndarray:
        """items: [(stream, frame index, has_prev)] for the batch resident at frames_dev, laid out as all current
        frames first, then the predecessors of those that have one, in the same order.  One detector + classifier
        pass over the current frames (faces[0] per frame, as the server does) and one forensic pass."""
        return self.local_records_waves(frames_dev, height, width, [items], conf_thr)[0]

    def local_records_groups(self, groups, conf_thr: float = 0.5) -> List[List[np.ndarray]]:
        """Several look-ahead groups - [(frames_dev, height, width, waves_items), ...] - each as `local_records_waves`, group
        k on worker handle k % len(workers), the groups of one round side by side on host threads (the library calls release
        the GIL; one device pass is a chain of dependent launches and stream waits, a second pass's kernels fill its gaps:
        DESIGN section 5).  A record is a pure function of its frame and the predecessor, so which handle computes a group
        does not matter; returns the groups' record blocks in the order given."""
        nw = len(self.workers)
        if nw == 1 or len(groups) == 1:
            return [self.local_records_waves(*g, conf_thr=conf_thr) for g in groups]
        import threading

        out: List = [None] * len(groups)
        errs: List = []

        def run(k0):
            try:
                for k in range(k0, len(groups), nw):
                    out[k] = self.local_records_waves(*groups[k], conf_thr=conf_thr, handle=self.workers[k0])
            except BaseException as e:                              # noqa: BLE001 - re-raised on the calling thread
                errs.append(e)

        th = [threading.Thread(target=run, args=(k0,)) for k0 in range(min(nw, len(groups)))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        return out

    def local_records_waves(self, frames_dev: int, height: int, width: int, waves_items, conf_thr: float = 0.5,
                            handle=None) -> List[np.ndarray]:
        """Several waves of this rank in ONE device pass (look-ahead batching: what a frame contributes to its record is a
        pure function of the frame and its predecessor, only the replay is sequential, so the frames of the next L waves can
        share a detector / classifier / forensic batch; the exchange stays one all-gather per wave, in wave order).
        waves_items: L lists of (stream, frame index, has_prev); device layout: the current frames of all waves, wave by
        wave, then the predecessors of those that have one in the same order.  Returns the L record blocks."""
        flat = [it for items in waves_items for it in items]
        m = len(flat)
        if any(len(items) > self.n_streams for items in waves_items):
            raise ValueError("more frames in a wave than record slots")
        prevs = [i for i, it in enumerate(flat) if it[2]]
        prev_index = np.full(m + len(prevs), -1, np.int32)
        prev_index[m:] = -2                                   # predecessor-only frames: a gray plane, no signals
        for k, i in enumerate(prevs):
            prev_index[i] = m + k
        dev = handle if handle is not None else self.h
        boxes, logits, _ = dev.analyze_batch_device(frames_dev, m, height, width, forced_boxes=None,
                                                    confidence_threshold=conf_thr, max_faces=1, with_forensics=False)
        scores, mdiff = dev.forensic_signals_device(frames_dev, m + len(prevs), height, width, prev_index)
        small = height < 30 or width < 30
        blocks, i = [], 0
        for items in waves_items:
            block = np.full((self.n_streams, RECORD_FLOATS), -1.0, np.float64)
            for j, (stream, frame, _) in enumerate(items):
                p = np.nan
                if boxes[i] and not small:
                    x, y, w, h = boxes[i][0]
                    lg = logits[i][0]
                    if not np.isnan(lg):
                        p = self.face_probability(lg, h, w)
                block[j] = (stream, frame, p, mdiff[i], *scores[i], len(boxes[i]))
                i += 1
            blocks.append(block)
        return blocks

    def finish_wave(self, block: np.ndarray) -> Dict[int, List[dict]]:
        """collective: exchange + replay; identical return value on every rank"""
        return replay_all(self.replicas, exchange(block, self.transport, self.h, self.group))

    def finish_waves(self, blocks: Sequence[np.ndarray]) -> List[Dict[int, List[dict]]]:
        """collective: the exchanges of consecutive waves (one all-gather per wave, batched into one library call on
        the RCCL transport), then the replays in wave order; identical return value on every rank"""
        return [replay_all(self.replicas, rec) for rec in exchange_waves(blocks, self.transport, self.h, self.group)]
