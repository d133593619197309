"""What build a measurement belongs to: a hash of the kernel sources (works on the GPU box, where the snapshot has no
.git) plus the commit id when git can tell it.  profiles/make_traffic.py stamps profiles/traffic*.json with it;
bench.py recomputes it and reports whether the PMC figure it quotes (`roofline.traffic`) was measured on the sources
it is running (a stale traffic.json went unnoticed)."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "real-time-video-deepfake-detection_amd", "csrc")


# the sources the classifier's launches are built from (what the PMC traffic figure describes); the detector / JPEG /
# forensic files can change without touching it
CLASSIFIER_SOURCES = ("b0_kernels.hip", "b0_kernels.h", "b0_plan.hip", "gemm_split.hip", "gemm_split_bf16.hip", "gemm_split_impl.h",
                      "kernel_util.h")


def kernel_sources_sha16():
    h = hashlib.sha256()
    for p in sorted(os.path.join(CSRC, n) for n in CLASSIFIER_SOURCES):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def commit():
    if os.environ.get("DFD_COMMIT"):
        return os.environ["DFD_COMMIT"]
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None if r.returncode == 0 else None
    except Exception:
        return None


def stamp():
    return {"kernel_sources_sha16": kernel_sources_sha16(), "commit": commit()}


if __name__ == "__main__":
    import json

    print(json.dumps(stamp()))
