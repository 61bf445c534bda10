import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # built artefacts are git-ignored: (re)build what is missing or stale, as __graft_entry__.build() does (hipcc cross-compiles
    # without a GPU).  Building is all that happens here; a failure surfaces in the tests that need the library.
    try:
        from fiducials_amd import build as fb

        fb.build(force=False)
        import oracle

        oracle.build()
    except Exception as e:  # noqa: BLE001
        print("conftest: build step failed:", e)
    # torch brings its own copy of the HIP runtime: when it initialises AFTER libfid_amd.so has touched the device it finds
    # "no HIP GPUs".  The tests that hand torch tensors to the library therefore initialise torch first, whatever the order
    # of the test files.
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception as e:  # noqa: BLE001
        print("conftest: torch.cuda.init failed:", e)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
