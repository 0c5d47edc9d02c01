import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a clean checkout has no built artefacts (they are git-ignored): build libaadg_hip.so (hipcc cross-compiles
    # without a GPU) and the oracle before collecting; both are no-ops when up to date
    try:
        from aadg_amd import build as _b
        _b.build_hip()
        from oracle import oracle as _o
        _o.build()
    except Exception as e:  # noqa: BLE001 -- tests that need the artefacts will fail loudly on their own
        print("conftest: build step failed: %s" % e)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip():
    """The HIP path: fails loudly (never skips) when the library or the GPU is missing."""
    import torch
    from aadg_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    return _lib
