import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def P():
    return importlib.import_module("pl-svo_amd")


@pytest.fixture(scope="session")
def ob():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def gpu_ctx(P):
    """A plsvo_ctx on cuda:0.  Fails (does not skip) when the HIP library or the device is missing:
    GPU tests must never pass on a silent fallback."""
    ctx = P.capi.Context(0)
    yield ctx
    ctx.close()
