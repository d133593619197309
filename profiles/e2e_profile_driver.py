"""Runs only the 1080p end-to-end extra of bench.py (for rocprofv3 --stats on that path)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rtdfd_amd  # noqa: E402

W = rtdfd_amd.weights
h = rtdfd_amd._lib.Handle(W.pack_all(W.seeded_state_dict(0), W.seeded_ssd_state_dict(0)), device=0, max_batch=64)
print(bench.e2e_frames(h, 0, None, 0, steps=int(os.environ.get("E2E_STEPS", "10"))))
h.close()
