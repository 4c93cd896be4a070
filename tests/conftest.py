import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built():
    """Native pieces compiled (library, oracle, emulation)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "tests", "emul")])
    lib = os.path.join(REPO, "tiny_renderer_amd", "lib", "libtiny_renderer.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "tiny_renderer_amd", "csrc"), "-j4"])
    return True


@pytest.fixture(scope="session")
def synthetic(built):
    import tiny_renderer_amd as T
    return T.synthetic_scene()


@pytest.fixture(scope="session")
def small_synthetic(built):
    import tiny_renderer_amd as T
    return T.synthetic_scene(n_lat=12, n_lon=24, tex_size=256)


def _assets(name):
    from tests import helpers as H
    a = H.load_assets_py(name)
    if a is None:
        pytest.skip("reference assets (%s) not available on this box" % name)
    return a


@pytest.fixture(scope="session")
def diablo(built):
    return _assets("diablo")


@pytest.fixture(scope="session")
def african_head(built):
    return _assets("african_head")
