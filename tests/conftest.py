import os
import sys

import numpy as np
import pytest

# torch bundles its own libamdhip64.so.7; if it is loaded AFTER the system one that
# libvoldor_hip.so links against, torch sees "No HIP GPUs". Importing torch first makes both
# share one HIP runtime (same soname). Only needed for tests that mix torch and the library.
if os.environ.get("VOLDOR_TESTS_NO_TORCH") != "1" and os.path.exists("/dev/kfd"):
    try:
        import torch  # noqa: F401
    except Exception:
        pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from voldor_amd import capi
        return capi.lib().vk_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.build()
    return o


@pytest.fixture(scope="session")
def small_scene():
    from voldor_amd import synth
    return synth.make_scene(w=160, h=120, n_flows=4, fx=80, fy=80, cx=80, cy=60, seed=7)


def K9(fx, fy, cx, cy):
    return np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], np.float32)


@pytest.fixture
def eight_point_bootstrap():
    """The fast mode's default two-view bootstrap is the five-point LMedS (round 6: the reference's estimator, voldor/geometry.cpp:316-326); the oracle and the
    reference-pipeline goldens (tests/golden/ref_*.npz, where the reference would call OpenCV) start from the 8-point pose.  A test that holds the EM loop of a
    fast window against them must start it from the same pose -- the bootstrap itself is held in tests/test_fivept.py.  Sets
    what `--bootstrap_points -1` stands for (vk_debug_switch "bootstrap_default") for the duration of the test."""
    import hooks
    prev = hooks.debug_switch("bootstrap_default", 8)
    yield
    hooks.debug_switch("bootstrap_default", prev)
