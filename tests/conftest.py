import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scalable-collaborative-zksnark_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """one zk_ctx on GPU 0; fails loudly if libzkhip.so or the GPU is missing (no fallback)"""
    import zkhip

    c = zkhip.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def co():
    """the plain-C CPU oracle (checker only)"""
    import coracle

    coracle.lib()
    return coracle
