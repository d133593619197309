"""CPU: the C-ABI library loads and exports every symbol include/dfd_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dfd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfd_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(pkg):
    assert _declared() == sorted(pkg._lib.SIGNATURES), "include/dfd_hip.h and _lib.SIGNATURES drifted"


def test_library_exports_every_symbol(pkg):
    if not os.path.exists(pkg._lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = pkg._lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.dfd_abi_version() == 1


def test_no_cpu_fallback(pkg):
    """Without a GPU, creating a handle must fail loudly rather than compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    blob = pkg.weights.pack_b0(pkg.weights.seeded_state_dict(0))
    with pytest.raises(pkg._lib.DfdError) as e:
        pkg._lib.Handle(blob, device=0, max_batch=1)
    assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "real-time-video-deepfake-detection_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
