import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "text-generation-inference_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    lib = os.path.join(PKG, "lib", "libtgis_hip.so")
    if not os.path.exists(lib):  # a checkout without built artefacts: compile once (hipcc is in the image)
        import __graft_entry__

        __graft_entry__.build()

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests(request):
    """GPU tests build whole models (one of them with the default KV pool of 85 % of the free memory); what a finished
    test leaves in the caching allocator must not starve the multi-process tests that follow in the same session."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        import torch

        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
