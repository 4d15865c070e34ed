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
def sk_ctx():
    """One device context for the whole GPU session.  No skip, no fallback: without the
    in-tree libsickle_amd.so or without a gfx950 device the gpu tests FAIL."""
    # torch first: its wheel bundles its own HIP runtime, and whichever copy of libamdhip64 a
    # process loads first is the one every later library gets.  bench.py has the same order.
    import torch
    torch.cuda.is_available()
    from sickle_amd import capi
    ctx = capi.Context(device=0, slots=2)
    yield ctx
    ctx.close()
