"""GPU: `python bench.py` keeps the driver's contract - exactly one JSON line on stdout with the fields the driver and
the judge read (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload, `roofline` and - in the full run only - `cpu_baseline`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-e2e",
                          "--no-streams", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 1e4 and abs(d["ms_per_step"] * d["value"] / 1e3 - 256) < 1.0      # value = batch / step time
    assert d["parity"]["max_abs_logit_err_vs_oracle"] <= 1e-3


def test_two_rank_launch_of_the_bench_rehearsed_on_one_gpu():
    """The driver starts `--gpus N` as `python -m torch.distributed.run ... bench.py --gpus N`.  With DFD_BENCH_REHEARSE=1
    two ranks share this box's GPU (torch.distributed over gloo, vote exchange on the documented torch fallback): the
    multi-rank control flow of the file - barriers, MAX-over-ranks timing, aggregate value, the frame-sharded streams
    with every frame verified against the single-GPU sequence - runs on real hardware before an 8-GPU node ever sees it.
    A rehearsal of control flow, not a measurement."""
    env = dict(os.environ, DFD_BENCH_REHEARSE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--no-e2e", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert abs(d["ms_per_step"] * d["value"] / 1e3 - 2 * 256) < 2.0                        # whole-job aggregate over both ranks
    c5 = d["config5"]
    assert c5["transport"] == "torch" and c5["verdicts_equal_single_gpu"] is True and c5["verified_frames_per_stream"] == 32 * 2      # every frame of every stream (32 waves x 2 ranks)
