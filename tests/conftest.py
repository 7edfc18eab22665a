import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


try:  # property tests draw the same examples on every run: a failure seen by the driver is one that reproduces here
    from hypothesis import settings as _hyp_settings

    _hyp_settings.register_profile("deterministic", derandomize=True, deadline=None, database=None)
    _hyp_settings.load_profile("deterministic")
except ImportError:  # hypothesis is optional; the property tests skip themselves without it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: every gpu-marked test skips (whether or not it goes through the `renderer` fixture)."""
    if not any("gpu" in it.keywords for it in items):
        return
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def liboxcull():
    """liboxcull.so loaded with prototypes; host-only entry points (mesh blob arithmetic) run without a GPU."""
    from oxylus_amd import lib as L

    L.build()
    return L.load()


@pytest.fixture(scope="session")
def renderer():
    """One RendererInstance (oxc_ctx) on cuda:0 for the whole GPU session."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oxylus_amd.renderer import RendererInstance

    r = RendererInstance(0)
    yield r
    r.close()
