"""pytest configuration: paths, the `gpu` marker, shared fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "template-speech-recognition_amd"), os.path.join(ROOT, "oracle"), ROOT,
          os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def ref():
    """The reference's compiled kernels (oracle/_ref) or None when absent."""
    import build_ref
    build_ref.build()
    return build_ref.load()
