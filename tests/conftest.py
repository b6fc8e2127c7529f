import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native library is a build artefact (git-ignored): make sure it exists before any test imports it
    try:
        import rgbd_amd  # noqa: F401
        from rgbd_amd import _lib

        if not os.path.exists(_lib._SO) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            _lib.build()
    except Exception as e:  # tests that need the library will report the real error
        print(f"[conftest] could not build librgbd_amd.so: {e}")


@pytest.fixture(scope="session")
def kat():
    return dict(np.load(os.path.join(GOLDEN, "coder_kat.npz")))


@pytest.fixture(scope="session")
def gc_tables(kat):
    from oracle import coder

    return coder.Tables(kat["gc_cdf"], kat["gc_sizes"], kat["gc_offsets"])


@pytest.fixture(scope="session")
def synth_sd():
    import rgbd_amd  # noqa: F401
    from rgbd_amd import synth

    return synth.synthetic_state_dict(0)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"model_{name}.npz")))


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress must fail, not sit on the GPU box: every `gpu` test without its own timeout gets
    one (method "thread": a wait inside the HIP runtime never returns to the interpreter, so a signal would not fire)."""
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(300, method="thread"))
