import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


@pytest.fixture(scope="session")
def pkg():
    import rtdfd_amd

    return rtdfd_amd


@pytest.fixture(scope="session")
def seeded_sd(pkg):
    return pkg.weights.seeded_state_dict(0)


@pytest.fixture(scope="session")
def ssd_sd(pkg):
    return pkg.weights.seeded_ssd_state_dict(0)


@pytest.fixture(scope="session")
def b0_handle(pkg, seeded_sd):
    """One classifier handle shared by the GPU tests (fails loudly if the .so or GPU is missing)."""
    blob = pkg.weights.pack_all(seeded_sd, pkg.weights.seeded_ssd_state_dict(0))
    h = pkg._lib.Handle(blob, device=0, max_batch=16)
    yield h
    h.close()


@pytest.fixture(scope="session")
def mtcnn_sd(pkg):
    return pkg.weights.seeded_mtcnn_state_dict(0)


@pytest.fixture(scope="session")
def mt_handle(pkg, seeded_sd, mtcnn_sd):
    """Classifier + detector + MTCNN cascade in one handle (the reference's full per-face path)."""
    blob = pkg.weights.pack_all(seeded_sd, pkg.weights.seeded_ssd_state_dict(0), mtcnn_sd)
    h = pkg._lib.Handle(blob, device=0, max_batch=64)       # predict classifies every detection of a frame
    yield h
    h.close()
