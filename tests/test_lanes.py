"""CPU: the host logic of `_lib.ClassifierLanes` (several classifier forwards in flight on one device) with stand-in
handles - lane rotation, the device-side ordering calls around a forward that must run alone, the shared tile table, the
untimed overlap check and its fallback to another stream-priority pool.  The kernels behind it have their own GPU tests
(tests/test_b0_batch256_gpu.py)."""
import pytest


class _FakeHandle:
    log = []

    def __init__(self, blob=b"", device=0, max_batch=8):
        self.name = f"h{len([e for e in _FakeHandle.log if e[0] == 'create'])}"
        _FakeHandle.log.append(("create", self.name))
        self.options, self.table, self.closed = {}, "", False

    def warmup(self, n, m):
        _FakeHandle.log.append(("warmup", self.name, n, m))
        if not self.table:
            self.table = f"tiles-of-{self.name}"

    def tiles_export(self):
        return self.table

    def tiles_import(self, text):
        self.table = text
        return 1

    def set_option(self, k, v):
        self.options[k] = v
        _FakeHandle.log.append(("option", self.name, k, v))

    def classify_device(self, x, n, y):
        _FakeHandle.log.append(("classify", self.name, y))

    def wait_for(self, other):
        _FakeHandle.log.append(("wait", self.name, other.name))

    def sync(self):
        _FakeHandle.log.append(("sync", self.name))

    def close(self):
        self.closed = True


class _Buf:
    def __init__(self, p):
        self.ptr = p


@pytest.fixture()
def lanes_mod(pkg, monkeypatch):
    _FakeHandle.log = []
    monkeypatch.setattr(pkg._lib, "Handle", _FakeHandle)
    return pkg._lib


def test_lanes_rotate_share_the_tile_table_and_close_only_what_they_made(lanes_mod):
    first = _FakeHandle()
    lanes = lanes_mod.ClassifierLanes(b"", lanes=3, first=first)
    assert len(lanes) == 3 and lanes.handles[0] is first
    lanes.warmup(256)
    assert [h.table for h in lanes.handles] == ["tiles-of-h0"] * 3            # lane 0 measures, the others import
    went = [lanes.submit(1, 256, 100 + i) for i in range(7)]
    assert went == [0, 1, 2, 0, 1, 2, 0]
    lanes.set_option("bf16_activations", 1)
    assert all(h.options["bf16_activations"] == 1 for h in lanes.handles)
    made = lanes.handles[1:]
    lanes.close()
    assert not first.closed and all(h.closed for h in made)
    assert lanes.handles == [first] and lanes.submit(1, 8, 200) == 0          # what is left is the caller's handle
    with pytest.raises(ValueError):
        lanes_mod.ClassifierLanes(b"", lanes=0)


def test_a_forward_that_runs_alone_is_ordered_on_the_device_both_ways(lanes_mod):
    lanes = lanes_mod.ClassifierLanes(b"", lanes=2)
    lanes.submit(1, 8, 10)                                                     # lane 0
    _FakeHandle.log.clear()
    k = lanes.submit_alone(1, 8, 11)                                           # lane 1: waits for lane 0, lane 0 then waits for it
    assert k == 1
    assert _FakeHandle.log == [("wait", "h1", "h0"), ("classify", "h1", 11), ("wait", "h0", "h1")]
    assert lanes.submit(1, 8, 12) == 0                                         # rotation continues
    assert not any(e[0] == "sync" for e in _FakeHandle.log)                    # no host wait anywhere


def test_overlap_check_moves_lane_1_only_when_the_lanes_are_not_faster(lanes_mod, monkeypatch):
    import time

    # a clock the test drives: every classify on a lane costs `cost[name]` ms of "GPU time", lanes in flight overlap
    lanes = lanes_mod.ClassifierLanes(b"", lanes=2)
    ys = [_Buf(1), _Buf(2)]
    now = [0.0]
    monkeypatch.setattr(time, "perf_counter", lambda: now[0])

    def install(step_ms_one, step_ms_two, step_ms_two_high):
        def classify(self, x, n, y):
            both = len({e[1] for e in _FakeHandle.log[-1:] if e[0] == "classify"} | {self.name}) > 1
            high = lanes.handles[1].options.get("stream_priority", 0) == 1
            now[0] += (step_ms_two_high if high else step_ms_two) * 1e-3 if both else step_ms_one * 1e-3
            _FakeHandle.log.append(("classify", self.name, y))
        monkeypatch.setattr(_FakeHandle, "classify_device", classify)

    install(3.0, 2.7, 2.7)                                                     # lanes 10 % faster: nothing to do
    rep = lanes.check_overlap(1, 8, ys, steps=6)
    assert rep["lane1_priority"] == 0 and "lanes_ms_high_priority" not in rep
    assert "stream_priority" not in lanes.handles[1].options
    install(3.0, 3.0, 2.7)                                                     # lanes in line; the other pool overlaps: moved
    rep = lanes.check_overlap(1, 8, ys, steps=6)
    assert rep["lane1_priority"] == 1 and lanes.handles[1].options["stream_priority"] == 1
    lanes.handles[1].options.pop("stream_priority")
    install(3.0, 3.0, 3.2)                                                     # the other pool is worse: put back
    rep = lanes.check_overlap(1, 8, ys, steps=6)
    assert rep["lane1_priority"] == 0 and lanes.handles[1].options["stream_priority"] == 0
