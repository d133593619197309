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
